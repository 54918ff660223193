R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_iteration.py -m gpu -q --timeout 600 -k "side_stream" 2>&1 | grep -v Warning | grep -B2 -A25 "Error\|assert" | head -120
