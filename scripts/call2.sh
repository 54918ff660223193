cd /root/repo
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/r02_sq_counters.txt
export AVC_LIB_NAME=libavc_p0.so
bash scripts/pmc_pass.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" r02_pmc_sdf_a -- python $GRAFT_REPO_ROOT/scripts/sdf_only_bench.py 4194304 sdffwd > /dev/null
bash scripts/pmc_pass.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" r02_pmc_sdf_b -- python $GRAFT_REPO_ROOT/scripts/sdf_only_bench.py 4194304 sdffwd > /dev/null
bash scripts/pmc_pass.sh "SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVES" r02_pmc_sdf_c -- python $GRAFT_REPO_ROOT/scripts/sdf_only_bench.py 4194304 sdffwd > /dev/null
cat gpurun_out/r02_sq_counters.txt; for t in a b c; do echo "== $t"; grep -A12 "mlp_sdf\|mlp_render" gpurun_out/r02_pmc_sdf_$t.txt | head -40; tail -3 gpurun_out/r02_pmc_sdf_$t.log; done
