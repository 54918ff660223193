R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt5
rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o kt -- python $R/scripts/train_time.py 5120 30 > /tmp/kt5.log 2>&1
grep "Runner.train" /tmp/kt5.log
python $R/scripts/rocpd_stats.py /tmp/kt5 14 2>&1 | cut -c1-150 | tee $R/gpurun_out/c51_train_5120.txt
python $R/scripts/rocpd_gaps.py /tmp/kt5 5 2>&1 | tail -8 | cut -c1-160 | tee -a $R/gpurun_out/c51_train_5120.txt
