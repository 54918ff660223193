R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in libavc.so libavc_attnvalu.so; do
  rm -rf /tmp/at_$v
  AVC_LIB_NAME=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/at_$v -o p -- python $R/scripts/attn_time.py 2 > /tmp/at.log 2>&1
  echo "== $v"; grep -h "attn" /tmp/at_$v/*kernel_stats.csv /tmp/at_$v/*/*kernel_stats.csv 2>/dev/null | cut -c1-200
done | tee $R/gpurun_out/c18_attn_kernels.txt
