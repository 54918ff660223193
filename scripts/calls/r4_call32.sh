#!/bin/bash
# loss tail kernel + in-kernel final reduction of the shade-loss sums: parity, iteration tests, timings, census
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_c32
timeout 1200 python -m pytest tests/test_gpu_glue.py tests/test_gpu_iteration.py tests/test_glue_golden.py -x -q -m gpu 2>&1 | tail -8
for i in 1 2; do
  timeout 300 python scripts/silhouette_time.py 7000 512 100 2>&1 | tail -1
  timeout 300 python bench.py --res 224 --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('224^2 ms/step', d['ms_per_step'])"
done 2>&1 | tee $R/gpurun_out/r4_c32/timing.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --res 224 --steps 6 --warmup 5 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
python $R/scripts/rocpd_census.py /tmp/kt $R/gpurun_out/r4_c32/seq_224.txt > $R/gpurun_out/r4_c32/census_224.txt 2>&1
cut -c1-130 $R/gpurun_out/r4_c32/census_224.txt | grep -v "^\[" | head -60
