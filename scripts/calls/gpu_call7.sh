R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 900 > gpurun_out/c7_tests_full.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/c7_tests_full.log | tail -20
cd /tmp
for v in "" _vnosplit; do
  rm -rf /tmp/kt$v; AVC_LIB_NAME=libavc$v.so rocprofv3 --kernel-trace --stats -d /tmp/kt$v -o kt -- python $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-extra > /tmp/kt$v.log 2>&1
  echo "lib$v"; python $R/scripts/rocpd_stats.py /tmp/kt$v 14 | cut -c1-150 | grep -E "vit_|upsample|composite|kernel  " 
  tail -1 /tmp/kt$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done > $R/gpurun_out/c7_vit.txt 2>&1
cat $R/gpurun_out/c7_vit.txt
cd $R; for i in 1 2; do timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step'])"; done
timeout 600 python bench.py --res 224 --steps 10 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['ms_per_step'], d['kernel_ms_per_step'])"
