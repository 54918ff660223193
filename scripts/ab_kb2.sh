#!/bin/bash
# A/B of kernel variants inside ONE gpurun call: per-kernel times at 4 Mi points (scripts/kb2.py) for libavc.so and every
# libavc_<name>.so given (built with scripts/build_variant.sh), two passes so that drift of the box shows.
#   gpurun -- 'bash scripts/ab_kb2.sh r5_call01 noeh norr'      -> gpurun_out/r5_call01.txt
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/$tag.txt
: > $out
for pass in 1 2; do
  for v in "" "$@"; do
    lib=libavc${v:+_$v}.so
    AVC_LIB_NAME=$lib timeout 300 python scripts/kb2.py ${NPTS:-4194304} 2>&1 | grep -v Warning | tail -1 >> $out
  done
done
cat $out
