#!/bin/bash
# fused residual blocks of the per-iteration CLIP call: parity, then timings with the switch on / off (alternating, twice)
mkdir -p gpurun_out/r4_c30
timeout 900 python -m pytest tests/test_gpu_clip.py -x -q -m gpu -s 2>&1 | grep -v Warning | tail -25 > gpurun_out/r4_c30/tests.txt
cat gpurun_out/r4_c30/tests.txt
for f in 1 0 1 0; do
  echo "== AVC_CLIP_FUSED_BLOCKS=$f" 
  AVC_CLIP_FUSED_BLOCKS=$f timeout 300 python scripts/silhouette_time.py 7000 512 100 2>&1 | tail -1
  AVC_CLIP_FUSED_BLOCKS=$f timeout 300 python bench.py --res 224 --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('224^2 ms/step', d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r4_c30/timing.txt
