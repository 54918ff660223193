#!/bin/bash
# second batch of head fusions (multi-workgroup column sums, pack memo, s_val, binary mask, the prior render in four launches): parity, timings
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_c40
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_smpl_prior.py tests/test_shapegen.py tests/test_gpu_iteration.py tests/test_gpu_glue.py tests/test_gpu_dataset_train.py -x -q -m gpu 2>&1 | grep -v Warning | tail -6
for i in 1 2; do
  timeout 300 python scripts/silhouette_time.py 7000 512 100 2>&1 | tail -1
  timeout 300 python bench.py --res 224 --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('224^2 ms/step', d['ms_per_step'])"
  timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('512^2 ms/step', d['ms_per_step'])"
done 2>&1 | tee $R/gpurun_out/r4_c40/timing.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --res 224 --steps 6 --warmup 5 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
python $R/scripts/rocpd_census.py /tmp/kt $R/gpurun_out/r4_c40/seq_224.txt > $R/gpurun_out/r4_c40/census_224.txt 2>&1
grep "one step\|^head\|^after\|colsum" $R/gpurun_out/r4_c40/census_224.txt | cut -c1-120
