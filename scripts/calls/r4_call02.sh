# round 4, call 2: ring parity after the stale-partials fix; consumers per product; how long do ring lines survive in L2 (experiment builds)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 180 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
timeout 600 python -m pytest tests/test_gpu_ring.py "tests/test_gpu_kernels.py::test_parameter_gradients_match_golden" -x -q 2>&1 | tail -8 | tee gpurun_out/r4_c02_tests.txt
timeout 300 python scripts/ring_bench.py 4194304 plain ring:2:3 ring:3:3 ring:4:3 ring:4:2 ring:5:2 2>&1 | grep -v Warn | tee gpurun_out/r4_c02_ring_4Mi.txt
AVC_LIB_NAME=libavc_expb1.so timeout 300 python scripts/ring_bench.py 4194304 ring:2:3 ring:2:2 ring:1:3 ring:3:2 2>&1 | grep -v Warn | tee gpurun_out/r4_c02_ring_4Mi_expb1.txt
AVC_LIB_NAME=libavc_expb2.so timeout 300 python scripts/ring_bench.py 4194304 ring:2:3 ring:2:2 ring:3:2 2>&1 | grep -v Warn | tee gpurun_out/r4_c02_ring_4Mi_expb2.txt
export RING_REPS=1
for c in FETCH_SIZE WRITE_SIZE; do
bash scripts/pmc_pass.sh "$c" r4_c02_pmc_${c}_base -- python $R/scripts/ring_bench.py 4194304 plain ring:4:3 > /dev/null
AVC_LIB_NAME=libavc_expb1.so bash scripts/pmc_pass.sh "$c" r4_c02_pmc_${c}_expb1 -- python $R/scripts/ring_bench.py 4194304 ring:2:3 > /dev/null
AVC_LIB_NAME=libavc_expb2.so bash scripts/pmc_pass.sh "$c" r4_c02_pmc_${c}_expb2 -- python $R/scripts/ring_bench.py 4194304 ring:2:3 > /dev/null
done
for f in gpurun_out/r4_c02_pmc_*.txt; do echo $f; grep -A1 "mlp_bwd\|weight_grad" $f | head -8; done
