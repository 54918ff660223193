R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
timeout 600 python -m pytest tests/test_gpu_clip.py tests/test_gpu_iteration.py -m gpu -q --timeout 600 -k "hip_graphs or small_nets or ablation" 2>&1 | tail -3
