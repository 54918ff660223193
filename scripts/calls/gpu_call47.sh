R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
timeout 900 python -m pytest tests/test_gpu_clip.py tests/test_gpu_iteration.py tests/test_gpu_dataset_train.py tests/test_gpu_parallel.py -m gpu -q --timeout 900 2>&1 | tail -4
( timeout 300 python scripts/silhouette_time.py 7000 512 30; AVC_CLIP_GRAPH=0 timeout 300 python scripts/silhouette_time.py 7000 512 30 ) 2>&1 | grep "silhouette mode" | tee gpurun_out/c47_silhouette.txt
for g in 1 0; do AVC_CLIP_GRAPH=$g timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph=$g', d['ms_per_step'], {k:v.get('ms_per_step') for k,v in d['extra_configs'].items()})"; done | tee -a gpurun_out/c47_silhouette.txt
