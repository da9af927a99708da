#!/bin/bash
for e in "X=1" "TRAMP_PRIOR_FUSE=0" "TRAMP_EW_TILE=0" "TRAMP_ROTATE_OUTPUT=0" "TRAMP_EW_SINGLE=0" "TRAMP_MFMA_SPLIT=8"; do
  echo "== $e"; env $e timeout 200 python tools/probes/diag_rank.py 2>&1 | grep -E "batch 3 damping None|rank" | cut -c1-160
done
