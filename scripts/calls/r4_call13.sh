# round 4, call 13: full kernel census of the silhouette-mode iteration (7000 rays) -- which small kernels make up the GPU-side tail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && mkdir -p $R/gpurun_out
AVC_PREFETCH_VIEW=0 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/scripts/silhouette_time.py 7000 512 30 > /tmp/kt.log 2>&1
python $R/scripts/rocpd_stats.py /tmp/kt 70 > $R/gpurun_out/r4_c13_silhouette_kernels.txt 2>&1
tail -3 /tmp/kt.log
head -75 $R/gpurun_out/r4_c13_silhouette_kernels.txt
