R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c16_bench.json 2> gpurun_out/c16_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c16_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['steps'], d['warmup'], d['roofline']['kernel'], round(d['roofline']['frac'],3), d['cpu_baseline']['value'], {k:v.get('ms_per_step') for k,v in d['extra_configs'].items()})
PY
