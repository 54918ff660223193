R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_clip.py tests/test_clip_score.py -m gpu -q --timeout 600 2>&1 | tail -3
( for v in libavc.so libavc_gnall.so libavc_gnocc1.so libavc_gemml1.so; do AVC_LIB_NAME=$v python scripts/score_bench.py 512 2>&1 | grep "B=" | sed "s/^/$v /"; done; AVC_VIT_LIBRARY_GEMM=1 python scripts/score_bench.py 512 | grep "B="
  for v in libavc.so libavc_gnall.so; do AVC_LIB_NAME=$v python scripts/score_bench.py 64 2>&1 | grep "B=" | sed "s/^/$v /"; done; AVC_VIT_LIBRARY_GEMM=1 python scripts/score_bench.py 64 | grep "B=" ) | tee gpurun_out/c24_score.txt
cd /tmp && export TMPDIR=/tmp
for v in libavc.so libavc_gnall.so; do
rm -rf /tmp/sb; AVC_LIB_NAME=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb -o p -- python $R/scripts/score_bench.py 512 > /tmp/sb.log 2>&1
f=$(ls /tmp/sb/*kernel_stats.csv /tmp/sb/*/*kernel_stats.csv 2>/dev/null | head -1)
echo "== $v"; head -9 $f | cut -c1-60,150-260
done | tee $R/gpurun_out/c24_score_kernels.txt
