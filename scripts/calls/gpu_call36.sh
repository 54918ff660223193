R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o p -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extra > /tmp/kt3.log 2>&1
f=$(ls /tmp/kt3/*kernel_stats.csv /tmp/kt3/*/*kernel_stats.csv 2>/dev/null | head -1)
grep -h "upsample" $f | cut -c1-160 | tee $R/gpurun_out/c36_upsample.txt
tail -1 /tmp/kt3.log | cut -c1-200
