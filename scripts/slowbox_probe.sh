#!/bin/bash
# Some boxes of the pool run mlp_bwd_kernel ~45 % slower than the others (14.2 instead of 9.5 ms per 4 Mi points; every other kernel normal;
# profiles/r05_ab_kernels.txt).  This probe times the base library and, ONLY on such a box, the cache-policy variants of the backward kernel,
# with the device's clocks / partition modes and the kernel's HBM counters beside them.   gpurun -- 'bash scripts/slowbox_probe.sh <tag> v1 v2 ..'
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/$tag.txt
python scripts/kb2.py 2>&1 | tail -1 > $out
bwd=$(sed 's/.*render_points_bwd \([0-9.]*\).*/\1/' $out)
echo "bwd $bwd" >> $out
if python -c "import sys; sys.exit(0 if float('$bwd') > 12.0 else 1)"; then
  echo "SLOW BOX" >> $out
  rocm-smi --showclocks --showmemorypartition --showcomputepartition --showpower --showperflevel 2>/dev/null | grep -v "^=\|^$" >> $out
  for v in "$@"; do AVC_LIB_NAME=libavc_$v.so python scripts/kb2.py 2>&1 | tail -1 >> $out; done
  python scripts/kb2.py 2>&1 | tail -1 >> $out
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
    d=$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/sb/$d -o p -- python $GRAFT_REPO_ROOT/scripts/kb2.py > /tmp/sb_$d.log 2>&1
  done
  python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/sb | grep -A8 "mlp_bwd" >> $GRAFT_REPO_ROOT/$out
fi
cat $GRAFT_REPO_ROOT/$out 2>/dev/null || cat $out
