#!/bin/bash
# end of round 4: full GPU suite, smoke(), the measurement pass on HEAD, silhouette-mode timings
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_c42
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $R/gpurun_out/r4_c42/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/call_profile.sh r04 2>&1 | tail -12
for n in 7000 12544; do timeout 300 python scripts/silhouette_time.py $n 512 100 2>&1 | tail -1; done | tee $R/gpurun_out/r04_silhouette_mode.txt
