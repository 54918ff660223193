R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 900 > gpurun_out/c6_tests_full.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/c6_tests_full.log | tail -20
( for bs in 1024 2048 4096; do echo "AVC_WG_BLOCKS_PER_SPLIT=$bs"; AVC_WG_BLOCKS_PER_SPLIT=$bs timeout 600 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step'])"; done ) > gpurun_out/c6_splits.txt 2>&1
cat gpurun_out/c6_splits.txt
timeout 600 python bench.py --res 224 --steps 8 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['ms_per_step'], d['kernel_ms_per_step'])"
