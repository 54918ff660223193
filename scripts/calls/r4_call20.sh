R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for h in 512 256; do echo "== dataset $h"; timeout 600 python scripts/makeview_opprof.py 7000 $h 2>&1 | grep -v "Warn\|amdgpu.ids\|WeightNorm\|WARNING\|warn"; done > gpurun_out/r4_c20_makeview.txt; head -100 gpurun_out/r4_c20_makeview.txt
