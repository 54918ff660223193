R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( timeout 300 python scripts/silhouette_time.py 7000 512 60 ) 2>&1 | grep "silhouette mode" | tee gpurun_out/c54_silhouette.txt
