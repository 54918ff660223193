R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_iteration.py tests/test_gpu_dataset_train.py tests/test_gpu_parallel.py -m gpu -q --timeout 900 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step'], {k:(v.get('ms_per_step'), v.get('kernel_ms_per_step',{}).get('avc_weight_grad(all pairs)')) for k,v in d['extra_configs'].items()})"
