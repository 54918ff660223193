R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_iteration.py -m gpu -q --timeout 600 -k "side_stream or silhouette" 2>&1 | tail -15
for f in 0 1 0 1; do echo "AVC_OVERLAP_HEAD=$f"; AVC_OVERLAP_HEAD=$f timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v.get('ms_per_step') for k,v in d['extra_configs'].items()})"; done | tee gpurun_out/c19_overlap.txt
