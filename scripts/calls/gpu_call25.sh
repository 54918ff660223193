R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
( for v in libavc.so libavc_gxcd.so libavc_gsmall.so libavc_gsmallx.so libavc_gkb1.so libavc.so; do AVC_LIB_NAME=$v python scripts/score_bench.py 512 2>&1 | grep "B=" | sed "s/^/$v /"; done
  for v in libavc.so libavc_gxcd.so libavc_gsmall.so libavc_gsmallx.so; do AVC_LIB_NAME=$v python scripts/score_bench.py 64 2>&1 | grep "B=" | sed "s/^/$v /"; done ) | tee gpurun_out/c25_score.txt
