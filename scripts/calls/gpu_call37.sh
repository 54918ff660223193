R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c37_bench.json 2> gpurun_out/c37_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c37_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['traffic_source'], {k:v.get('ms_per_step') for k,v in d['extra_configs'].items()})
PY
