R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python scripts/silhouette_opprof.py 7000 2>&1 | grep -v "Warn\|amdgpu.ids\|WeightNorm\|WARNING" > gpurun_out/r4_c19_opprof.txt; head -80 gpurun_out/r4_c19_opprof.txt
