R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt6
rocprofv3 --kernel-trace --stats -d /tmp/kt6 -o kt -- python $R/scripts/silhouette_time.py 7000 512 30 > /tmp/kt6.log 2>&1
grep "silhouette mode" /tmp/kt6.log
python $R/scripts/rocpd_stats.py /tmp/kt6 10 2>&1 | cut -c1-140 | tee $R/gpurun_out/c57_silhouette_trace.txt
python $R/scripts/rocpd_gaps.py /tmp/kt6 5 2>&1 | grep "steady" | tee -a $R/gpurun_out/c57_silhouette_trace.txt
