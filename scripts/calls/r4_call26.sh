R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_iteration.py -q -x -k "small_nets" 2>&1 | grep -v "Warning:\|amdgpu.ids\|WeightNorm\|WARNING:root" | tail -40
AVC_FUSED_HEAD=0 timeout 900 python -m pytest tests/test_gpu_iteration.py -q -x -k "small_nets" 2>&1 | tail -3
