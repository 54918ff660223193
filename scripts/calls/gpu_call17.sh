R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_clip.py tests/test_clip_score.py -m gpu -q --timeout 600 2>&1 | tail -15
for v in libavc.so libavc_attnvalu.so; do AVC_LIB_NAME=$v timeout 120 python scripts/attn_time.py 2 512 2>&1 | grep "B="; done | tee gpurun_out/c17_attn.txt
