#!/bin/bash
# one slab by default (full-size tests), packed attention forward: lane-pair exchange vs 2-byte stores
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_c38
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_clip.py -x -q -m gpu 2>&1 | tail -2
for lib in libavc.so libavc_pd.so libavc.so libavc_pd.so; do
  echo "== $lib"
  AVC_LIB_NAME=$lib timeout 300 python scripts/silhouette_time.py 7000 512 100 2>&1 | tail -1
done 2>&1 | tee $R/gpurun_out/r4_c38/timing.txt
cd /tmp && export TMPDIR=/tmp
for lib in libavc.so libavc_pd.so; do
  rm -rf /tmp/kt
  AVC_LIB_NAME=$lib rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --res 224 --steps 6 --warmup 5 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
  echo "== $lib"; python $R/scripts/rocpd_stats.py /tmp/kt 60 | grep "attn" | cut -c1-150
done 2>&1 | tee -a $R/gpurun_out/r4_c38/timing.txt
