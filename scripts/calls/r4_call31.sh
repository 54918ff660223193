#!/bin/bash
# fused residual blocks after the row-per-wave LayerNorm kernels: unit tests, A/B timings, launch census of a 224^2 step
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_c31
timeout 600 python -m pytest tests/test_gpu_clip.py -x -q -m gpu -k "fused or layernorm or batched_scoring" 2>&1 | tail -3
for f in 1 0 1 0; do
  echo "== AVC_CLIP_FUSED_BLOCKS=$f"
  AVC_CLIP_FUSED_BLOCKS=$f timeout 300 python scripts/silhouette_time.py 7000 512 100 2>&1 | tail -1
  AVC_CLIP_FUSED_BLOCKS=$f timeout 300 python bench.py --res 224 --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('224^2 ms/step', d['ms_per_step'])"
done 2>&1 | tee $R/gpurun_out/r4_c31/timing.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --res 224 --steps 6 --warmup 5 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
python $R/scripts/rocpd_census.py /tmp/kt $R/gpurun_out/r4_c31/seq_224.txt > $R/gpurun_out/r4_c31/census_224.txt 2>&1
cut -c1-130 $R/gpurun_out/r4_c31/census_224.txt
