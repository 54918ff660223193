#!/bin/bash
# build avatarclip_amd/libavc_<name>.so with extra hipcc flags (A/B timing inside ONE gpurun call: AVC_LIB_NAME=libavc_<name>.so python scripts/kb2.py)
#   scripts/build_variant.sh nt "-DWG_DMA_AUX=2"
set -e
name=$1; shift
cd "$(dirname "$0")/.."
AVC_LIB_NAME=libavc_$name.so AVC_OBJ_SUFFIX=_$name AVC_EXTRA_FLAGS="$*" python -c "from avatarclip_amd import build; print(build.build())"
