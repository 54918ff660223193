# round 4, call 15: the measurement pass on HEAD (kernel trace + census, FETCH / WRITE PMC passes, the bench line)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && git rev-parse --short HEAD > gpurun_out/.commit 2>/dev/null
bash scripts/call_profile.sh r04 2>&1 | tail -30
