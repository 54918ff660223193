R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_glue.py -x -q 2>&1 | grep -v "Warning:\|amdgpu.ids\|WeightNorm\|WARNING:root" | tail -30
