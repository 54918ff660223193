R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
( for v in libavc_gsmallx.so libavc_gs4a.so libavc_gs4b.so libavc_gs4c.so libavc_gs3d.so libavc_gsmallx.so; do AVC_LIB_NAME=$v python scripts/score_bench.py 512 2>&1 | grep "B=" | sed "s/^/$v /"; done
  for v in libavc_gsmallx.so libavc_gs4a.so libavc_gs4b.so libavc_gs4c.so; do AVC_LIB_NAME=$v python scripts/score_bench.py 64 2>&1 | grep "B=" | sed "s/^/$v /"; done ) | tee gpurun_out/c26_score.txt
