R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
echo "== health"; timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y, torch.cuda.get_device_name(0))
PY
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 0
echo "== kernels"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -x 2>&1 | tail -2
echo "== bench"; timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | cut -c1-200
