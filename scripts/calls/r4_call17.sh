# round 4, call 17: split-K granularity of the weight-gradient launch at the silhouette mode's sizes (448 K / 800 K points)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for b in 256 128 64; do
for m in 7000 12544; do
echo -n "blocks/split $b: "; AVC_WG_BLOCKS_PER_SPLIT=$b timeout 300 python scripts/silhouette_time.py $m 512 60 2>&1 | grep "silhouette mode"
done; done | tee gpurun_out/r4_c17_wgsplit.txt
for b in 256 128; do echo -n "blocks/split $b: "; AVC_WG_BLOCKS_PER_SPLIT=$b timeout 300 python bench.py --res 224 --steps 10 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('224^2 ms/step', d['ms_per_step'], d['kernel_ms_per_step'])"; done | tee -a gpurun_out/r4_c17_wgsplit.txt
