R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4
( python scripts/score_bench.py 512; AVC_VIT_LIBRARY_GEMM=1 python scripts/score_bench.py 512; python scripts/score_bench.py 64; AVC_VIT_LIBRARY_GEMM=1 python scripts/score_bench.py 64; python scripts/score_bench.py 8; AVC_VIT_LIBRARY_GEMM=1 python scripts/score_bench.py 8 ) 2>&1 | grep "B=" | tee gpurun_out/c27_score.txt
