R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
timeout 300 python scripts/silhouette_sync_debug.py 7000 2>&1 | grep -v "Warning\|WARNING\|amdgpu\|WeightNorm" | tail -22 | tee gpurun_out/c53_sync.txt
