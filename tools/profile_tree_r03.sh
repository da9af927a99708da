#!/bin/bash
# Tree-structured EP at image scale (tools/bench_tree.py) on the own N-D FFT engine and on rocFFT, plus the kernel trace of
# the 4096 x 4096 x 4 sparse-gradient case (the round-1 profile of the same command had rocFFT at 48 % of kernel time).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_tree_r03
mkdir -p $OUT
for eng in own rocfft; do
  export TRAMP_CONV_FFT=$eng
  echo "== engine $eng" >> $OUT/bench_tree.txt
  python3 tools/bench_tree.py 1024 4 f64 sgrad >> $OUT/bench_tree.txt 2>&1
  python3 tools/bench_tree.py 4096 4 f64 sgrad >> $OUT/bench_tree.txt 2>&1
  python3 tools/bench_tree.py 4096 4 f32 tv >> $OUT/bench_tree.txt 2>&1
done
export TRAMP_CONV_FFT=own
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_tree.py 4096 4 f64 sgrad > $OUT/trace.out 2>&1
cd $REPO
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats_tree_sgrad_4096x4096_x4.csv
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/bench_tree.txt
head -14 $OUT/kernel_stats_tree_sgrad_4096x4096_x4.csv | cut -c1-150
