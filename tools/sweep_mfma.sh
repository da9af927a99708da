#!/bin/bash
# knob sweep of the batch-64 f64 passes at the BASELINE shape
N=${1:-100000}
run() { echo "== $*"; env "$@" timeout 150 python tools/bench_matvec.py $N 64 2>&1 | grep -E "NT\(Vt\)|NN\(Vt\)"; }
run X=1
run TRAMP_MFMA_F64_SHAPE=4
run TRAMP_MFMA_F64_SHAPE=4 TRAMP_MFMA_RING=1
run TRAMP_MFMA_SPLIT=40
run TRAMP_MFMA_SPLIT=64
run TRAMP_NT_RT=1
