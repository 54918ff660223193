#!/bin/bash
# register / spill / LDS usage of every kernel of one HIP source (development aid):  scripts/kres.sh avc_mlp_bwd.hip [extra hipcc flags]
src=$1; shift
cd "$(dirname "$0")/../avatarclip_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Rpass-analysis=kernel-resource-usage "$@" -c $src -o /tmp/kres_$$.o 2>&1 \
 | sed -n 's/.*remark: [^ ]* *//p' | sed 's/ \[-Rpass.*//' \
 | awk '/Name/{name=$3} /VGPRs:/{v=$2} /AGPRs/{a=$2} /ScratchSize/{s=$3} /SGPRs Spill/{ss=$3} /VGPRs Spill/{vs=$3} /Occupancy/{o=$3} /LDS Size/{printf "%-110s VGPR %3s AGPR %3s vspill %3s sspill %3s scratch %5s occ %s lds %s\n", substr(name,1,110), v, a, vs, ss, s, o, $4}'
rm -f /tmp/kres_$$.o
