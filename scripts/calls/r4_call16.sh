# round 4, call 16: SDF kernel with the activation evaluated in f16 after the operand conversion: time + parity
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
for lib in libavc.so libavc_f16act.so libavc.so libavc_f16act.so; do
AVC_LIB_NAME=$lib timeout 300 python scripts/kb2.py 4194304 2>&1 | grep "npts" | cut -c1-120
done | tee gpurun_out/r4_c16_f16act.txt
for lib in libavc.so libavc_f16act.so; do
echo "== parity with $lib" | tee -a gpurun_out/r4_c16_f16act.txt
AVC_LIB_NAME=$lib timeout 600 python -m pytest tests/test_gpu_kernels.py -q -s -k "sdf_and_point_forward or upsample_steps or full_sampling_chain or render_matches_golden" 2>&1 | grep -i "sdf\|passed\|failed\|err\|median\|max" | head -40 | tee -a gpurun_out/r4_c16_f16act.txt
done
