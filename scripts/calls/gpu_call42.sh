R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
for b in 1024 512 392 256 1024 392; do echo "AVC_WG_BLOCKS_PER_SPLIT=$b"; AVC_WG_BLOCKS_PER_SPLIT=$b timeout 300 python bench.py --res 224 --steps 10 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step'])"; done | tee gpurun_out/c42_wgsplit_224.txt
