#!/bin/bash
# up-sampling with four rays per wavefront: parity, kernel time at 224^2 and 512^2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_c41
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_iteration.py tests/test_gpu_dataset_train.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -v Warning | tail -4
timeout 300 python scripts/silhouette_time.py 7000 512 100 2>&1 | tail -1
timeout 300 python bench.py --res 224 --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('224^2 ms/step', d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
python $R/scripts/rocpd_stats.py /tmp/kt 14 | cut -c1-140 | tee $R/gpurun_out/r4_c41/stats_512.txt
tail -1 /tmp/kt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('512^2 ms/step (tracer)', d['ms_per_step'])"
