R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_iteration.py -q 2>&1 | tail -3
for m in 7000 12544; do
timeout 300 python scripts/silhouette_time.py $m 512 60 2>&1 | grep "silhouette mode"
AVC_PREFETCH_VIEW=0 timeout 300 python scripts/silhouette_time.py $m 512 60 2>&1 | grep "silhouette mode"
done | tee gpurun_out/r4_c12_silhouette.txt
timeout 600 python scripts/silhouette_hostprof.py 7000 512 2>&1 | grep -v "Warning:\|amdgpu.ids\|WeightNorm\|WARNING:root" > gpurun_out/r4_c12_hostprof.txt; head -3 gpurun_out/r4_c12_hostprof.txt
