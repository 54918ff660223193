R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python scripts/bitcheck_head.py 2>&1 | grep "^silhouettes"
