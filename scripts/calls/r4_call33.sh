#!/bin/bash
# rasteriser in rounds of 1024 faces, fused split sums + un-packing of the dense gradient: parity, timings (split cap A/B), census
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_c33
timeout 1500 python -m pytest tests/test_gpu_ring.py tests/test_smpl_prior.py tests/test_gpu_iteration.py tests/test_gpu_kernels.py -x -q -m gpu -k "fused_split or raster or iteration or prior or grad" 2>&1 | tail -4
for ms in 256 128 64; do
  echo "== AVC_WG_MAX_SPLITS=$ms"
  AVC_WG_MAX_SPLITS=$ms timeout 300 python scripts/silhouette_time.py 7000 512 100 2>&1 | tail -1
  AVC_WG_MAX_SPLITS=$ms timeout 300 python bench.py --res 224 --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('224^2 ms/step', d['ms_per_step'])"
  AVC_WG_MAX_SPLITS=$ms timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('512^2 ms/step', d['ms_per_step'])"
done 2>&1 | tee $R/gpurun_out/r4_c33/timing.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --res 224 --steps 6 --warmup 5 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
python $R/scripts/rocpd_census.py /tmp/kt $R/gpurun_out/r4_c33/seq_224.txt > $R/gpurun_out/r4_c33/census_224.txt 2>&1
cut -c1-130 $R/gpurun_out/r4_c33/census_224.txt | grep -v "^\[" | head -24
