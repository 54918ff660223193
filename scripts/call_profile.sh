# Measurement pass of the round (run through gpurun): kernel trace, PMC traffic of the MLP kernels, the bench line.   usage: call_profile.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcs /tmp/kt && mkdir -p /tmp/pmcs $R/gpurun_out
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
python $R/scripts/rocpd_stats.py /tmp/kt 30 > $R/gpurun_out/${T}_bench_512_kernel_trace_stats.txt 2>&1
python $R/scripts/rocpd_gaps.py /tmp/kt 5 | tail -15 >> $R/gpurun_out/${T}_bench_512_kernel_trace_stats.txt 2>&1
python $R/scripts/rocpd_census.py /tmp/kt > $R/gpurun_out/${T}_step_census.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs/$c -o p -- python $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-extra > /tmp/pmc_$c.log 2>&1
done
python $R/scripts/pmc_summary.py /tmp/pmcs --json $R/gpurun_out/${T}_pmc_traffic.json "$(cat $R/gpurun_out/.commit 2>/dev/null || cat $R/.commit_id 2>/dev/null || echo unknown)" "gpurun MI355X box" 4 512 64 > $R/gpurun_out/${T}_pmc_fetch_write.txt 2>&1
cp $R/gpurun_out/${T}_pmc_traffic.json $R/profiles/ 2>/dev/null   # for the bench run below on THIS box; back home copy gpurun_out/${T}_* into profiles/ (only gpurun_out/ is merged back)
cd $R && python bench.py > gpurun_out/${T}_bench_512.json 2> gpurun_out/${T}_bench_512.err
tail -c 4000 gpurun_out/${T}_bench_512.json; echo; head -12 gpurun_out/${T}_bench_512_kernel_trace_stats.txt; tail -6 gpurun_out/${T}_pmc_fetch_write.txt
