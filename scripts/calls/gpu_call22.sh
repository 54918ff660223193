R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python $R/scripts/score_bench.py 512 2>&1 | grep "B="
rm -rf /tmp/sb; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb -o p -- python $R/scripts/score_bench.py 512 > /tmp/sb.log 2>&1
f=$(ls /tmp/sb/*kernel_stats.csv /tmp/sb/*/*kernel_stats.csv 2>/dev/null | head -1)
head -25 $f | cut -c1-230 | tee $R/gpurun_out/c22_score_kernels.txt
