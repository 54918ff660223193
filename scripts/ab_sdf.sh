#!/bin/bash
# round 6: the SDF-only kernel with 32 vs 64 points per wavefront (AVC_SDF_POINTS_PER_WAVE) for libavc.so and every libavc_<name>.so given,
# two passes, then the value comparison of the two kernels (same arithmetic per point: equal bit for bit is the expectation).
#   gpurun -- 'bash scripts/ab_sdf.sh r6_call02 [variant ...]'      -> gpurun_out/r6_call02.txt
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/$tag.txt
: > $out
for pass in 1 2; do
  for v in "" "$@"; do
    lib=libavc${v:+_$v}.so
    for ppw in 32 64; do
      AVC_LIB_NAME=$lib AVC_SDF_POINTS_PER_WAVE=$ppw timeout 300 python scripts/sdf_ab.py 2>&1 | grep -v Warning | tail -1 >> $out
    done
  done
done
python - >> $out <<'P'
import torch
for R, S in ((512 * 512, 32), (512 * 512, 8), (65536, 64), (1000, 7)):
    a, b = torch.load("/tmp/sdf_ab_32_%d_%d.pt" % (R, S)), torch.load("/tmp/sdf_ab_64_%d_%d.pt" % (R, S))
    print("values 32 vs 64 points per wave, %d x %d: equal %s, max |diff| %.3e" % (R, S, bool(torch.equal(a, b)), (a - b).abs().max().item()))
P
cat $out
