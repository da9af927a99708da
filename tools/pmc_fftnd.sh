#!/bin/bash
# HBM traffic of the N-D FFT passes and the N-D spectral kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, no
# other trace domain): python3 tools/bench_fftnd.py f64 3 one conv 4096x4096 4 (one ConvChannel message pair per repetition).
# Algorithmic bytes of one pass over a batch of 4 packed 4096 x 4096 images: 4 x 8 388 608 complex x 16 B = 537 MB read + 537 MB
# written.  FETCH_SIZE is reported x 2 on gfx950 (MI355X_MICROARCH.md); units KiB.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_fftnd
mkdir -p $OUT
export TRAMP_CONV_FFT=own
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $REPO/tools/bench_fftnd.py f64 3 one conv 4096x4096 4 > $OUT/$c.out 2>&1
done
cd $REPO
for c in FETCH_SIZE WRITE_SIZE; do echo "== $c"; python3 tools/pmc_parse.py $OUT/$c fft_; python3 tools/pmc_parse.py $OUT/$c spectral; done > $OUT/pmc_fftnd_traffic.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +8M -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/pmc_fftnd_traffic.txt
