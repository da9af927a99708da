#!/bin/bash
# Round-6 profiles (run on the GPU box through gpurun): rocprofv3 kernel-trace stats of the DEFAULT bench command, and
# separate --pmc passes over the componentwise updates added this round (tools/pmc_ew_r6.py) and over the existing ones
# (tools/pmc_ew.py: the redo of flagged instances changed).  Never --pmc together with a trace domain other than --kernel-trace.
# Usage: tools/profile_r06.sh [stats|pmc|all]
set -u
WHAT=${1:-all}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r06
mkdir -p $OUT
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  cd /tmp && export TMPDIR=/tmp
  # The set-up (untimed) runs on the rounds 2-5 eigensolver under the tracer: the default tridiagonal path is 250 000 launches at
  # k = 50 000, which rocprofv3's kernel trace does not get through in a quarter of an hour (tools/.. CHANGELOG 12.7), and
  # rocprofv3 (ROCm 7.2) segfaults inside rocSOLVER's syevd for k >= 24576: TRAMP_EIG_METHOD=split TRAMP_EIG_CAP=16384 keeps
  # every library eigensolve below that (the spectral split recurses once more).  The timed region is the default command's.
  TRAMP_EIG_METHOD=split TRAMP_EIG_CAP=16384 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
  tail -3 $OUT/stats.err
  cd $REPO
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  for script in pmc_ew_r6 pmc_ew; do
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${script}_fetch -- python3 tools/$script.py > $OUT/${script}_fetch.out 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${script}_write -- python3 tools/$script.py > $OUT/${script}_write.out 2>&1
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/${script}_valu -- python3 tools/$script.py > $OUT/${script}_valu.out 2>&1
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/${script}_waits -- python3 tools/$script.py > $OUT/${script}_waits.out 2>&1
  done
fi
python3 tools/summarize_r06.py $OUT $OUT/summary
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
