R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for fh in 1 0; do echo "== AVC_FUSED_HEAD=$fh"; AVC_TEST_VERBOSE=1 AVC_FUSED_HEAD=$fh timeout 900 python -m pytest tests/test_gpu_iteration.py -q -s -k "small_nets" 2>&1 | grep "rel \|iter \|passed\|failed"; done
