R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_clip.py tests/test_clip_score.py tests/test_clip_text.py -q 2>&1 | tail -3
for lib in libavc.so libavc_packx.so; do
echo "== $lib"
AVC_LIB_NAME=$lib timeout 300 python scripts/clip_graph_probe.py 2>&1 | grep "eager\|graphs"
AVC_LIB_NAME=$lib AVC_PREFETCH_VIEW=0 timeout 300 python scripts/silhouette_time.py 7000 512 60 2>&1 | grep "silhouette mode"
AVC_LIB_NAME=$lib timeout 300 python bench.py --res 224 --steps 10 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('224^2 ms/step', d['ms_per_step'])"
done | tee gpurun_out/r4_c14_directx.txt
