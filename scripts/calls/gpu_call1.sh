# round-3 GPU call 1: tests, wgrad / engine A-B at 4 Mi points, the L2-ring micro-benchmark (+ PMC), slab-size sweep, the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -80 ) > gpurun_out/c1_tests.log
grep -E "passed|failed|error" gpurun_out/c1_tests.log | tail -3
( for v in "" _wgstatic _wgnt _wgstaticnt _tr2 _v6 _v14 _pair; do AVC_LIB_NAME=libavc$v.so timeout 300 python scripts/kb2.py 4194304 2>&1 | tail -1; done ) > gpurun_out/c1_kb2.txt
cat gpurun_out/c1_kb2.txt
timeout 300 python scripts/ubench/run3.py > gpurun_out/r03_ubench3.txt 2>&1
cat gpurun_out/r03_ubench3.txt
cd /tmp && rm -rf /tmp/ub3 && for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/ub3/$c -o p -- python $R/scripts/ubench/run3.py > /tmp/ub3_$c.log 2>&1
done
python $R/scripts/pmc_summary.py /tmp/ub3 > $R/gpurun_out/r03_ubench3_pmc.txt 2>&1
grep -A3 "ub3_" $R/gpurun_out/r03_ubench3_pmc.txt | head -40
cd $R
( for sb in 65536 131072 262144 524288; do echo "AVC_SLAB_BLOCKS=$sb"; AVC_SLAB_BLOCKS=$sb timeout 600 python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step'])"; done ) > gpurun_out/c1_slabs.txt 2>&1
cat gpurun_out/c1_slabs.txt
timeout 900 python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
tail -c 2500 gpurun_out/c1_bench.json; tail -5 gpurun_out/c1_bench.err
