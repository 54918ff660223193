R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_clip.py -m gpu -q --timeout 600 -s -k batched_scoring 2>&1 | grep -i "packed vs\|assert\|Error" | head -20
