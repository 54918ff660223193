#!/bin/bash
# silhouette mode: granularity of the weight-gradient launch's K-split at ~14 k blocks
for b in 256 128 64 32; do
  echo "== AVC_WG_BLOCKS_PER_SPLIT=$b"
  AVC_WG_BLOCKS_PER_SPLIT=$b timeout 300 python scripts/silhouette_time.py 7000 512 150 2>&1 | tail -1
done
