R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_clip_score.py tests/test_gpu_clip.py -m gpu -q --timeout 600 -x 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -2 | cut -c1-300
