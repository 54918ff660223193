R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_clip.py tests/test_clip_score.py -m gpu -q --timeout 600 2>&1 | tail -8
( python scripts/score_bench.py 512; AVC_LIB_NAME=libavc_gemml1.so python scripts/score_bench.py 512; AVC_VIT_LIBRARY_GEMM=1 python scripts/score_bench.py 512; python scripts/score_bench.py 64; AVC_LIB_NAME=libavc_gemml1.so python scripts/score_bench.py 64; AVC_VIT_LIBRARY_GEMM=1 python scripts/score_bench.py 64 ) 2>&1 | grep "B=" | tee gpurun_out/c23_score.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sb; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb -o p -- python $R/scripts/score_bench.py 512 > /tmp/sb.log 2>&1
f=$(ls /tmp/sb/*kernel_stats.csv /tmp/sb/*/*kernel_stats.csv 2>/dev/null | head -1)
head -12 $f | cut -c1-200 | tee $R/gpurun_out/c23_score_kernels.txt
