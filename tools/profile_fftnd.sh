#!/bin/bash
# Kernel trace of the 2-D gradient / convolution channels on the own FFT engine and on rocFFT (run through gpurun).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_fftnd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for eng in own rocfft; do
  export TRAMP_CONV_FFT=$eng
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$eng -- python3 $REPO/tools/bench_fftnd.py f64 5 one > $OUT/$eng.out 2>&1
  f=$(find $OUT/$eng -name "*kernel_stats.csv" | head -1)
  echo "== $eng"; cat $OUT/$eng.out | tail -3
  head -25 "$f" | cut -c1-220
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
