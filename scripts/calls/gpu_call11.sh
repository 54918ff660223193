R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
export TMPDIR=/tmp
( for v in "" _b1 _b2 _b3; do AVC_LIB_NAME=libavc$v.so timeout 300 python scripts/kb2.py 4194304 2>&1 | tail -1; done ) > gpurun_out/c11_kb2.txt
cat gpurun_out/c11_kb2.txt
