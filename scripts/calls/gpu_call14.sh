R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
( for o in sorted layout sorted layout; do echo "pair order $o"; AVC_WG_PAIR_ORDER=$o timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step'])"; done ) > gpurun_out/c14_order.txt 2>&1
cat gpurun_out/c14_order.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 900 -k "parameter_gradients" 2>&1 | tail -2
