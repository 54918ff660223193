R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-extra > /tmp/kt.log 2>&1
tail -2 /tmp/kt.log | cut -c1-300
python $R/scripts/rocpd_census.py /tmp/kt $R/gpurun_out/r03_step_sequence.txt > $R/gpurun_out/r03_step_census.txt 2>&1
cat $R/gpurun_out/r03_step_census.txt
cd $R && timeout 900 python -m pytest tests/test_smpl_prior.py tests/test_gpu_kernels.py -m gpu -q --timeout 900 -k "shapegen or parameter_gradients" 2>&1 | tail -4
