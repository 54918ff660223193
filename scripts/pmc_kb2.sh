# FETCH_SIZE / WRITE_SIZE (+ L2 hit counters) of every MLP kernel at 4 Mi points (scripts/kb2.py), separate passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && mkdir -p /tmp/pk
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  d=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pk/$d -o p -- python $R/scripts/kb2.py > /tmp/pk_$d.log 2>&1
done
mkdir -p $R/gpurun_out; python $R/scripts/pmc_summary.py /tmp/pk | grep -A7 "mlp_\|weight_grad" > $R/gpurun_out/r02_pmc_kernels_4Mi.txt; cat $R/gpurun_out/r02_pmc_kernels_4Mi.txt
