R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_clip.py tests/test_clip_score.py -m gpu -q --timeout 600 -s 2>&1 | grep -v Warning | tail -12
( python scripts/score_bench.py 512; AVC_VIT_PACKED=0 python scripts/score_bench.py 512; python scripts/score_bench.py 64;  AVC_VIT_PACKED=0 python scripts/score_bench.py 64; python scripts/score_bench.py 8;  AVC_VIT_PACKED=0 python scripts/score_bench.py 8 ) 2>&1 | grep "B=" | tee gpurun_out/c28_score.txt
