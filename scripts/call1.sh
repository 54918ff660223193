cd /root/repo
python scripts/ubench/run2.py > gpurun_out/r02_ubench2.txt 2>&1
for v in p0 p1 p1v5 p0; do echo "=== $v" ; AVC_LIB_NAME=libavc_$v.so python scripts/kbench.py 4194304 2>&1 | grep npts; done > gpurun_out/r02_kbench_pair.txt 2>&1
tail -50 gpurun_out/r02_ubench2.txt; cat gpurun_out/r02_kbench_pair.txt
