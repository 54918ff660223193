# round 4, call 8: 8-rank readiness on one GPU (gloo, all ranks on device 0)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_parallel.py -x -q 2>&1 | grep -v "Warning:\|amdgpu.ids\|WeightNorm" | tail -60 > gpurun_out/r4_c08_parallel.txt; grep -n "Error\|error\|assert" gpurun_out/r4_c08_parallel.txt | head -20
