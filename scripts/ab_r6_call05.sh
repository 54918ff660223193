#!/bin/bash
# round 6, call 05: what does the weight DMA cost the MLP kernels?  chunk order rotated per workgroup, half the bytes, every chunk from one
# 1-KiB source; per-kernel times at 4 Mi points + the SDF kernel at the sampler's sizes, two passes
out=gpurun_out/r6_call05_dma_probes.txt; mkdir -p gpurun_out; : > $out
for pass in 1 2; do
  for v in "" rot rotnt half samesrc nodma; do
    lib=libavc${v:+_$v}.so
    AVC_LIB_NAME=$lib timeout 300 python scripts/kb2.py 4194304 2>&1 | grep -v Warning | tail -1 >> $out
    AVC_LIB_NAME=$lib timeout 300 python scripts/sdf_ab.py 2>&1 | grep -v Warning | tail -1 | cut -c1-120 >> $out
  done
done
cat $out
