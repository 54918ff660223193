# round-3 GPU call 2: full test log, fwd-kernel variants, effective clock per kernel, the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 900 > gpurun_out/c2_tests_full.log 2>&1
grep -E "^(PASSED|FAILED|ERROR)|passed|failed" gpurun_out/c2_tests_full.log | tail -70
grep -E "rel [0-9.e+-]+ cos|OUTLIER|Adam displacement|RCCL|torso-band|loss hip|worst per-tensor" gpurun_out/c2_tests_full.log | head -120
( for v in "" _fnopair _ftr2; do AVC_LIB_NAME=libavc$v.so timeout 300 python scripts/kb2.py 4194304 2>&1 | tail -1; done ) > gpurun_out/c2_kb2.txt
cat gpurun_out/c2_kb2.txt
cd /tmp && rm -rf /tmp/clk && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk -o p -- python $R/scripts/kb2.py 4194304 > /tmp/clk.log 2>&1
python $R/scripts/eff_clock.py /tmp/clk > $R/gpurun_out/r03_eff_clock.txt 2>&1; cat $R/gpurun_out/r03_eff_clock.txt
cd $R
timeout 900 python bench.py > gpurun_out/c2_bench.json 2> gpurun_out/c2_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c2_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['step_mfma_frac'])
print({k:(v.get('ms_per_step'), v.get('peak_hbm_gib'), v.get('rays_per_slab')) for k,v in d['extra_configs'].items()})
print(d['cpu_baseline']['value'])
PY
tail -3 gpurun_out/c2_bench.err
