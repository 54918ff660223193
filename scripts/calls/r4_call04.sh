# round 4, call 4: operand panels in uncached / fine-grained memory (streaming data kept out of the L2?) -- plain pair + ring
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for k in uncached finegrained; do
PANEL_ALLOC=$k timeout 300 python scripts/ring_bench.py 4194304 plain ring:2:3 ring:4:3 2>&1 | grep -v Warn | tee gpurun_out/r4_c04_ring_4Mi_$k.txt
PANEL_ALLOC=$k AVC_LIB_NAME=libavc_expb1.so timeout 300 python scripts/ring_bench.py 4194304 ring:2:3 ring:4:3 2>&1 | grep -v Warn | tee gpurun_out/r4_c04_ring_4Mi_${k}_expb1.txt
done
export RING_REPS=1
for c in FETCH_SIZE WRITE_SIZE; do
PANEL_ALLOC=uncached bash scripts/pmc_pass.sh "$c" r4_c04_pmc_${c}_uc -- python $R/scripts/ring_bench.py 4194304 plain ring:4:3 > /dev/null
PANEL_ALLOC=uncached AVC_LIB_NAME=libavc_expb1.so bash scripts/pmc_pass.sh "$c" r4_c04_pmc_${c}_uc_expb1 -- python $R/scripts/ring_bench.py 4194304 ring:4:3 > /dev/null
done
for f in gpurun_out/r4_c04_pmc_*.txt; do echo $f; grep -A1 "mlp_bwd\|weight_grad" $f | head -8; done
