# round 4, call 6: whole GPU suite after: d_sdf hi/lo tile (no fp32 override of the sdf bias), CLIP graph in-flight guard + pinned workspace,
# the 512^2 gradient-vs-oracle test, the falsifiable outlier-gradient variant
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "Warning:\|amdgpu.ids\|WeightNorm" | tail -150 > gpurun_out/r4_c06_tests.txt
tail -40 gpurun_out/r4_c06_tests.txt
