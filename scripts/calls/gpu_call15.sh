R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_clip.py tests/test_clip_score.py tests/test_clip_text.py -m gpu -q --timeout 900 2>&1 | tail -3
( python scripts/score_bench.py 512; AVC_VIT_LIBRARY_GEMM=1 python scripts/score_bench.py 512; python scripts/score_bench.py 64; AVC_VIT_LIBRARY_GEMM=1 python scripts/score_bench.py 64 ) 2>&1 | grep "B=" > gpurun_out/r03_score_bench.txt
cat gpurun_out/r03_score_bench.txt
