R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
( timeout 300 python scripts/train_time.py 512 200; timeout 300 python scripts/train_time.py 5120 100 ) 2>&1 | grep "Runner.train" | tee gpurun_out/c50_train_time.txt
