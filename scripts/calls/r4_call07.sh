# round 4, call 7: ShapeGen chain (BASELINE config 5's AvatarGen half) on the GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_shapegen.py -x -q -s 2>&1 | grep -v "Warning:\|amdgpu.ids\|WeightNorm" | tail -30 | tee gpurun_out/r4_c07_shapegen.txt
timeout 900 python scripts/pipeline_config5.py /tmp/c5 2>&1 | grep -v "Warning:\|amdgpu.ids\|WeightNorm" | tail -12 | tee -a gpurun_out/r4_c07_shapegen.txt
