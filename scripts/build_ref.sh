#!/bin/bash
# build libavc_ref.so from the committed HEAD (A/B timing against the working tree inside ONE gpurun call: boxes differ by up to 1.5x)
set -e
R=/root/repo; W=/tmp/avc_ref_tree
rm -rf $W; git -C $R worktree prune; git -C $R worktree add -f --detach $W HEAD > /dev/null 2>&1
(cd $W && python -c "from avatarclip_amd import build; build.build()")
cp $W/avatarclip_amd/libavc.so $R/avatarclip_amd/libavc_ref.so
git -C $R worktree remove --force $W
echo built libavc_ref.so from $(git -C $R rev-parse --short HEAD)
