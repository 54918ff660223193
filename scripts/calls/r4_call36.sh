#!/bin/bash
# hybrid rasteriser (face-parallel z-buffer + tile pass for large faces): parity incl. close-ups, CLIP tests, kernel times in the 512^2 bench
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_c36
timeout 900 python -m pytest tests/test_smpl_prior.py tests/test_shapegen.py tests/test_gpu_clip.py -x -q -m gpu -s 2>&1 | grep -v Warning | grep "eye\|MeshPrior\|passed\|failed\|Error\|assert" | tee $R/gpurun_out/r4_c36/tests.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
python $R/scripts/rocpd_stats.py /tmp/kt 40 | grep "raster\|calls" | tee $R/gpurun_out/r4_c36/raster_512.txt
tail -1 /tmp/kt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('512^2 ms/step (under the tracer)', d['ms_per_step'])"
