# round 4, call 1: first run of the role-specialised backward (ring) -- parity vs the panel path, timing, counters
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 180 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
timeout 600 python -m pytest tests/test_gpu_ring.py "tests/test_gpu_kernels.py::test_parameter_gradients_match_golden" -x -q 2>&1 | tail -15 | tee gpurun_out/r4_c01_tests.txt
timeout 300 python scripts/ring_bench.py 4194304 plain ring:2:6 ring:2:3 ring:3:4 ring:1:8 ring:2:12 2>&1 | tee gpurun_out/r4_c01_ring_4Mi.txt
timeout 300 python scripts/ring_bench.py 16777216 plain ring:2:6 2>&1 | tee gpurun_out/r4_c01_ring_16Mi.txt
export RING_REPS=1
bash scripts/pmc_pass.sh "FETCH_SIZE" r4_c01_pmc_fetch -- python $R/scripts/ring_bench.py 4194304 plain ring:2:6 > /dev/null
bash scripts/pmc_pass.sh "WRITE_SIZE" r4_c01_pmc_write -- python $R/scripts/ring_bench.py 4194304 plain ring:2:6 > /dev/null
grep -A2 "mlp_bwd\|weight_grad" gpurun_out/r4_c01_pmc_fetch.txt | head -20
grep -A2 "mlp_bwd\|weight_grad" gpurun_out/r4_c01_pmc_write.txt | head -20
