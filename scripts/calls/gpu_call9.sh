R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 900 > gpurun_out/c9_tests_full.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/c9_tests_full.log | tail -10
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/call_profile.sh r03 2>&1 | tail -30
cd /tmp && rm -rf /tmp/kt2 && rocprofv3 --kernel-trace -d /tmp/kt2 -o kt -- python $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-extra > /tmp/kt2.log 2>&1
python $R/scripts/rocpd_census.py /tmp/kt2 $R/gpurun_out/r03_step_sequence.txt > $R/gpurun_out/r03_step_census.txt 2>&1; head -12 $R/gpurun_out/r03_step_census.txt
