#!/bin/bash
# round 6, call 04: rotating DMA turns (AVC_DMA_TURNS) on every MLP kernel, the DMA-free upper bound, and the 4-wave (one wavefront per SIMD,
# up to 512 registers) builds of the backward and the training forward -- per-kernel times at 4 Mi points, two passes
out=gpurun_out/r6_call04_dma_turns.txt; mkdir -p gpurun_out; : > $out
for pass in 1 2; do
  for v in "" noturn nodma; do
    lib=libavc${v:+_$v}.so
    AVC_LIB_NAME=$lib timeout 300 python scripts/kb2.py 4194304 2>&1 | grep -v Warning | tail -1 >> $out
    AVC_LIB_NAME=$lib AVC_SDF_POINTS_PER_WAVE=32 timeout 300 python scripts/sdf_ab.py 2>&1 | grep -v Warning | tail -1 | cut -c1-190 >> $out
    AVC_LIB_NAME=$lib AVC_SDF_POINTS_PER_WAVE=64 timeout 300 python scripts/sdf_ab.py 2>&1 | grep -v Warning | tail -1 | cut -c1-190 >> $out
  done
  AVC_LIB_NAME=libavc_bwd4.so KB_MAX_BWD_WAVES=1024 timeout 300 python scripts/kb2.py 4194304 2>&1 | grep -v Warning | tail -1 >> $out
  AVC_LIB_NAME=libavc_fwd4.so KB_MAX_FWD_WAVES=1024 timeout 300 python scripts/kb2.py 4194304 2>&1 | grep -v Warning | tail -1 >> $out
done
cat $out
