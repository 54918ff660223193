# round 4, call 5: the record run of the ring experiment at the headline size (512^2 x 64 spp = 16.8 Mi points, two slabs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 600 python scripts/ring_bench.py 16777216 plain ring:4:3 ring:3:3 2>&1 | grep -v "Warn\|amdgpu.ids" | tee gpurun_out/r4_c05_ring_16Mi.txt
AVC_LIB_NAME=libavc_expb1.so timeout 600 python scripts/ring_bench.py 16777216 ring:6:3 2>&1 | grep -v "Warn\|amdgpu.ids" | tee gpurun_out/r4_c05_ring_16Mi_expb1.txt
export RING_REPS=1
for c in FETCH_SIZE WRITE_SIZE; do
bash scripts/pmc_pass.sh "$c" r4_c05_pmc_${c}_16Mi -- python $R/scripts/ring_bench.py 16777216 plain ring:4:3 > /dev/null
AVC_LIB_NAME=libavc_expb1.so bash scripts/pmc_pass.sh "$c" r4_c05_pmc_${c}_16Mi_expb1 -- python $R/scripts/ring_bench.py 16777216 ring:6:3 > /dev/null
done
for f in gpurun_out/r4_c05_pmc_*.txt; do echo $f; grep -A1 "mlp_bwd\|weight_grad" $f | head -8; done
