# round 6: kernel trace of the reference's DEFAULT mode (silhouette rays, 7 000 per iteration): what a whole-iteration HIP graph could save
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kts
python $R/scripts/silhouette_time.py 7000 512 60 > $R/gpurun_out/r06_silhouette_mode.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/kts -o kt -- python $R/scripts/silhouette_time.py 7000 512 40 > /tmp/kts.log 2>&1
python $R/scripts/rocpd_stats.py /tmp/kts 16 >> $R/gpurun_out/r06_silhouette_mode.txt 2>&1
python $R/scripts/rocpd_gaps.py /tmp/kts 5 | tail -12 >> $R/gpurun_out/r06_silhouette_mode.txt 2>&1
python $R/scripts/rocpd_census.py /tmp/kts >> $R/gpurun_out/r06_silhouette_mode.txt 2>&1
grep -v Warning $R/gpurun_out/r06_silhouette_mode.txt | tail -60
