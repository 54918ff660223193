R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_clip.py tests/test_clip_score.py tests/test_clip_text.py tests/test_gpu_iteration.py -m gpu -q --timeout 900 2>&1 | tail -3
cd /tmp
for v in "" _at4; do
  rm -rf /tmp/kt$v; AVC_LIB_NAME=libavc$v.so rocprofv3 --kernel-trace --stats -d /tmp/kt$v -o kt -- python $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-extra > /tmp/kt$v.log 2>&1
  echo "lib$v"; python $R/scripts/rocpd_stats.py /tmp/kt$v 14 | cut -c1-150 | grep -E "vit_|kernel  "
done > $R/gpurun_out/c8_vit.txt 2>&1
cat $R/gpurun_out/c8_vit.txt
