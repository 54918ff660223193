# round 4, call 22: fused glue kernels (shading + scatter + loss terms, resize + normalise): whole-iteration parity + timings A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_iteration.py tests/test_gpu_parallel.py tests/test_shapegen.py -q 2>&1 | tail -3
for fg in 1 0 1 0; do
echo "== AVC_FUSED_GLUE=$fg"
for m in 7000 12544; do AVC_FUSED_GLUE=$fg timeout 300 python scripts/silhouette_time.py $m 512 60 2>&1 | grep "silhouette mode"; done
AVC_FUSED_GLUE=$fg timeout 300 python bench.py --res 224 --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('224^2 ms/step', d['ms_per_step'])"
done | tee gpurun_out/r4_c22_fusedglue.txt
for fg in 1 0; do AVC_FUSED_GLUE=$fg timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512^2 ms/step', d['ms_per_step'])"; done | tee -a gpurun_out/r4_c22_fusedglue.txt
