# round 4, call 24: number of K-splits of the weight-gradient launch (the split buffers are summed afterwards: 256 x 2.6 MB per slab)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for ms in 256 128 64 256 128; do
echo -n "max splits $ms: "; AVC_WG_MAX_SPLITS=$ms timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512^2 ms/step %.2f' % d['ms_per_step'], {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done | tee gpurun_out/r4_c24_maxsplits.txt
for ms in 256 128 64; do
echo -n "max splits $ms: "; AVC_WG_MAX_SPLITS=$ms timeout 300 python bench.py --res 224 --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('224^2 ms/step %.2f' % d['ms_per_step'], {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done | tee -a gpurun_out/r4_c24_maxsplits.txt
