#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile_bench.sh <N> <steps> <tag>
set -u
N=${1:-32768}; STEPS=${2:-10}; TAG=${3:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# rocprofv3 (ROCm 7.2) segfaults inside rocSOLVER's syevd for k >= 24576, so the factorisation runs
# un-profiled first and the profiled command - the default bench otherwise - loads its result.
FDIR=/tmp/tramp_factors_$N
python3 bench.py --size $N --steps 2 --warmup 1 --no-cpu-baseline --no-denoiser --save-factors $FDIR > $OUT/bench_save.json 2> $OUT/save.err
ls -la $FDIR
ARGS="bench.py --size $N --steps $STEPS --warmup 3 --load-factors $FDIR"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
rm -rf $FDIR
# counter passes: own runs, kernel-trace only, on the kernels alone (rocSOLVER crashes under --pmc)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_kernels.py $N 64 > $OUT/fetch.out 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_kernels.py $N 64 > $OUT/write.out 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only the small files
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
du -sh $OUT
