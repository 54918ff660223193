R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_clip.py tests/test_clip_score.py tests/test_smpl_prior.py -m gpu -q --timeout 600 -s 2>&1 | grep -i "packed vs\|passed\|failed" | head -20
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sb; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb -o p -- python $R/scripts/score_bench.py 512 > /tmp/sb.log 2>&1
f=$(ls /tmp/sb/*kernel_stats.csv /tmp/sb/*/*kernel_stats.csv 2>/dev/null | head -1)
head -14 $f | cut -c1-75,150-250 | tee $R/gpurun_out/c30_score_kernels.txt
