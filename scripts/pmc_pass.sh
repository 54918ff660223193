#!/bin/bash
# usage: scripts/pmc_pass.sh "<counters>" <tag> -- <command...>   (run on the GPU box; output summary to gpurun_out/<tag>.txt)
ctr="$1"; tag="$2"; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > /tmp/pmc_$tag.log 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc_$tag > $R/gpurun_out/$tag.txt 2>&1
tail -c 1500 /tmp/pmc_$tag.log > $R/gpurun_out/$tag.log
head -c 6000 $R/gpurun_out/$tag.txt
