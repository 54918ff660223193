R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
bash scripts/pmc_pass.sh "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" r03_pmc_sdf_a -- python $R/scripts/sdf_only_bench.py 4194304 sdf,fwd > /dev/null
bash scripts/pmc_pass.sh "SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" r03_pmc_sdf_b -- python $R/scripts/sdf_only_bench.py 4194304 sdf,fwd > /dev/null
bash scripts/pmc_pass.sh "GRBM_GUI_ACTIVE" r03_pmc_sdf_clk -- python $R/scripts/sdf_only_bench.py 4194304 sdf,fwd > /dev/null
grep -A9 "mlp_sdf_kernel\|mlp_render_kernel" gpurun_out/r03_pmc_sdf_a.txt | head -24
grep -A9 "mlp_sdf_kernel\|mlp_render_kernel" gpurun_out/r03_pmc_sdf_b.txt | head -24
grep -A2 "mlp_sdf_kernel\|mlp_render_kernel" gpurun_out/r03_pmc_sdf_clk.txt | head -8
