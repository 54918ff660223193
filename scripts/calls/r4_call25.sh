# round 4, call 25: fused head (rays + near/far + prior resample in one launch, camera frame on the host, chess background in one launch)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_glue.py tests/test_smpl_prior.py tests/test_gpu_iteration.py tests/test_gpu_dataset_train.py -q 2>&1 | tail -5
for fh in 1 0 1 0; do
echo "== AVC_FUSED_HEAD=$fh"
for m in 7000 12544; do AVC_FUSED_HEAD=$fh timeout 300 python scripts/silhouette_time.py $m 512 60 2>&1 | grep "silhouette mode"; done
AVC_FUSED_HEAD=$fh timeout 300 python bench.py --res 224 --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('224^2 ms/step', d['ms_per_step'])"
done | tee gpurun_out/r4_c25_fusedhead.txt
