#!/bin/bash
# round 6, call 06: one wavefront per SIMD (4-wave workgroups, up to 512 registers) for the two training kernels, with and without their
# self re-reads (timing ablations: the upper bound of ANY register-resident scheme), per-kernel times at 4 Mi points, two passes
out=gpurun_out/r6_call06_wave512.txt; mkdir -p gpurun_out; : > $out
for pass in 1 2; do
  for v in "" fwdnorr bwdnorr; do
    AVC_LIB_NAME=libavc${v:+_$v}.so timeout 300 python scripts/kb2.py 4194304 2>&1 | grep -v Warning | tail -1 >> $out
  done
  for v in fwd4 fwd4norr; do
    AVC_LIB_NAME=libavc_$v.so KB_MAX_FWD_WAVES=1024 timeout 300 python scripts/kb2.py 4194304 2>&1 | grep -v Warning | tail -1 >> $out
  done
  for v in bwd4 bwd4norr; do
    AVC_LIB_NAME=libavc_$v.so KB_MAX_BWD_WAVES=1024 timeout 300 python scripts/kb2.py 4194304 2>&1 | grep -v Warning | tail -1 >> $out
  done
done
cat $out
