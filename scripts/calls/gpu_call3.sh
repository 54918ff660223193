R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_iteration.py tests/test_gpu_fullsize.py tests/test_smpl_prior.py -m gpu -q -rA --timeout 900 > gpurun_out/c3_tests_full.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/c3_tests_full.log | tail -30
grep -E "rel [0-9.e+-]+ cos|Adam displacement|torso-band|loss hip|worst per-tensor|max [0-9.e+-]+ mean|rel [0-9.e+-]+$" gpurun_out/c3_tests_full.log | head -150
( for v in "" _fnopair _ftr2; do AVC_LIB_NAME=libavc$v.so timeout 300 python scripts/kb2.py 4194304 2>&1 | tail -1; done ) > gpurun_out/c3_kb2.txt
cat gpurun_out/c3_kb2.txt
