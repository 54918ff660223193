#!/bin/bash
# full GPU suite, then the measurement pass of the round on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_c35
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $R/gpurun_out/r4_c35/tests.txt
bash scripts/call_profile.sh r04 2>&1 | tail -40
