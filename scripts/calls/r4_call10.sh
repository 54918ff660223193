R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for k in 1 2 3; do timeout 1700 python -m pytest tests/test_gpu_parallel.py -q 2>&1 | tail -2; done | tee gpurun_out/r4_c10_parallel.txt
