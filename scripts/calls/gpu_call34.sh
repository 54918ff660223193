R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
( for v in "" _v2 _v3 _v4 "" _v2 _v3 _v4; do AVC_LIB_NAME=libavc$v.so timeout 300 python scripts/kb2.py 4194304 2>&1 | tail -1; done ) > gpurun_out/c34_kb2.txt
cat gpurun_out/c34_kb2.txt
