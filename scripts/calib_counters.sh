# FETCH_SIZE / WRITE_SIZE calibration against a copy of known size (scripts/ubench: run2.py copy): 2 launches each of nt=0, nt=1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cal && mkdir -p /tmp/cal
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/cal/$c -o p -- python $R/scripts/ubench/run2.py copy > /tmp/cal_$c.log 2>&1
done
cat /tmp/cal_WRITE_SIZE.log | tail -2
python $R/scripts/pmc_summary.py /tmp/cal | grep -A2 ub2_copy
