R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python scripts/silhouette_hostprof.py 7000 512 2>&1 | grep -v "Warning:\|amdgpu.ids\|WeightNorm\|WARNING:root" > gpurun_out/r4_c11_hostprof.txt; head -5 gpurun_out/r4_c11_hostprof.txt
timeout 300 python scripts/silhouette_time.py 12544 512 40 2>&1 | grep "silhouette mode"
