# round 4, call 3: does the ring stay in L2 when the consumers are NOT the bottleneck (many consumers, all operands from the ring)?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
AVC_LIB_NAME=libavc_expb1.so timeout 300 python scripts/ring_bench.py 4194304 ring:4:3 ring:6:3 ring:8:3 ring:6:2 ring:6:6 2>&1 | grep -v Warn | tee gpurun_out/r4_c03_ring_4Mi_expb1.txt
AVC_LIB_NAME=libavc_expb1a0.so timeout 300 python scripts/ring_bench.py 4194304 ring:6:3 2>&1 | grep -v Warn | tee gpurun_out/r4_c03_ring_4Mi_expb1a0.txt
export RING_REPS=1
for c in FETCH_SIZE WRITE_SIZE; do
AVC_LIB_NAME=libavc_expb1.so bash scripts/pmc_pass.sh "$c" r4_c03_pmc_${c}_expb1_c6 -- python $R/scripts/ring_bench.py 4194304 ring:6:3 > /dev/null
AVC_LIB_NAME=libavc_expb1.so bash scripts/pmc_pass.sh "$c" r4_c03_pmc_${c}_expb1_c8 -- python $R/scripts/ring_bench.py 4194304 ring:8:3 > /dev/null
AVC_LIB_NAME=libavc_expb1a0.so bash scripts/pmc_pass.sh "$c" r4_c03_pmc_${c}_expb1a0_c6 -- python $R/scripts/ring_bench.py 4194304 ring:6:3 > /dev/null
done
for f in gpurun_out/r4_c03_pmc_*.txt; do echo $f; grep -A1 "mlp_bwd" $f | head -4; done
