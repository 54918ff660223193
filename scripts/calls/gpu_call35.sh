R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2
rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt -- python $R/bench.py --res 224 --steps 8 --warmup 5 --no-cpu-baseline --no-extra > /tmp/kt2.log 2>&1
tail -2 /tmp/kt2.log | cut -c1-300
python $R/scripts/rocpd_gaps.py /tmp/kt2 5 | tail -12
python $R/scripts/rocpd_census.py /tmp/kt2 | head -60 | cut -c1-110 > $R/gpurun_out/c35_census_224.txt
head -3 $R/gpurun_out/c35_census_224.txt
