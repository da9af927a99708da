#!/bin/bash
# split-K sweep of the MFMA passes at the BASELINE configs[3] shape (tail effects of the workgroup count)
N=${1:-100000}
for s in 8 11 12 15 16 24 32 48; do
  echo "== TRAMP_MFMA_SPLIT=$s"
  TRAMP_MFMA_SPLIT=$s timeout 120 python tools/bench_matvec.py $N 64 2>&1 | grep -E "NT\(Vt\)|NN\(Vt\)"
done
echo "== TRAMP_NT_RT=2"; TRAMP_NT_RT=2 timeout 120 python tools/bench_matvec.py $N 64 2>&1 | grep -E "NT\(Vt\)"
