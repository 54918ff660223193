R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
( for v in "" _spsplit "" ; do AVC_LIB_NAME=libavc$v.so timeout 300 python scripts/kb2.py 4194304 2>&1 | tail -1; done ) > gpurun_out/c33_kb2.txt
cat gpurun_out/c33_kb2.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_iteration.py -m gpu -q --timeout 900 2>&1 | tail -4
