#!/bin/bash
# round 6: SQ / LDS counters of the SDF kernel with and without its weight DMA (libavc.so, libavc_half.so, libavc_nodma.so), 4 Mi points
for v in "" half nodma; do
  lib=libavc${v:+_$v}.so
  AVC_LIB_NAME=$lib bash scripts/pmc_pass.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" r6_pmc_dma_a_${v:-base} -- python /root/repo/scripts/sdf_only_bench.py 4194304 sdf > /dev/null 2>&1
  AVC_LIB_NAME=$lib bash scripts/pmc_pass.sh "SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" r6_pmc_dma_b_${v:-base} -- python /root/repo/scripts/sdf_only_bench.py 4194304 sdf > /dev/null 2>&1
done
for f in gpurun_out/r6_pmc_dma_*.txt; do echo "== $f"; grep -A12 "mlp_sdf" $f | head -14; done
