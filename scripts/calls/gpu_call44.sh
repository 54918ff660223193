R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout 120 python - <<'PY' || { echo "HEALTH CHECK FAILED (box, not repo code)"; exit 0; }
import torch
x = torch.randn(4096, 4096, device="cuda"); y = (x @ x).sum().item(); print("torch matmul ok", y == y)
PY
( timeout 300 python scripts/silhouette_time.py 7000 512 30; timeout 300 python scripts/silhouette_time.py 50000 512 20 ) 2>&1 | grep "silhouette mode" | tee gpurun_out/c44_silhouette.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt4
rocprofv3 --kernel-trace --stats -d /tmp/kt4 -o kt -- python $R/scripts/silhouette_time.py 7000 512 12 > /tmp/kt4.log 2>&1
python $R/scripts/rocpd_gaps.py /tmp/kt4 5 | tail -16 | tee -a $R/gpurun_out/c44_silhouette.txt
