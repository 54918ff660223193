# round 4, call 23: launch census of one 224^2 step after the glue fusion (what is left of the tail)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --res 224 --steps 6 --warmup 5 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
python $R/scripts/rocpd_census.py /tmp/kt $R/gpurun_out/r4_c23_seq_224.txt > $R/gpurun_out/r4_c23_census_224.txt 2>&1
cat $R/gpurun_out/r4_c23_census_224.txt | cut -c1-130
