#!/bin/bash
# rasteriser with the round's faces staged through LDS: parity + kernel time
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4_c34
timeout 900 python -m pytest tests/test_smpl_prior.py tests/test_shapegen.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python scripts/silhouette_time.py 7000 512 100 2>&1 | tail -1
timeout 300 python scripts/silhouette_time.py 7000 512 100 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --res 224 --steps 6 --warmup 5 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
python $R/scripts/rocpd_census.py /tmp/kt $R/gpurun_out/r4_c34/seq_224.txt > $R/gpurun_out/r4_c34/census_224.txt 2>&1
cut -c1-130 $R/gpurun_out/r4_c34/census_224.txt | grep -v "^\[" | head -12
