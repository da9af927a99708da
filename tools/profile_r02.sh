#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): rocprofv3 kernel-trace stats of the DEFAULT bench command
# and separate --pmc passes over the kernels alone.  Usage: tools/profile_r02.sh [stats|pmc_mat|pmc_ew|all]
set -u
WHAT=${1:-all}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  # rocprofv3 (ROCm 7.2) segfaults inside rocSOLVER's syevd for k >= 24576: TRAMP_EIG_CAP=16384 keeps every
  # library eigensolve below that (the spectral split simply recurses once more); everything else is the default.
  TRAMP_EIG_CAP=16384 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
  tail -3 $OUT/stats.err
fi
if [ "$WHAT" = pmc_mat ] || [ "$WHAT" = all ]; then
  cd $REPO
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_mat -- python3 tools/pmc_kernels.py 100000 64 > $OUT/fetch_mat.out 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_mat -- python3 tools/pmc_kernels.py 100000 64 > $OUT/write_mat.out 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_mfma_mat -- python3 tools/pmc_kernels.py 100000 64 > $OUT/mfma_mat.out 2>&1
fi
if [ "$WHAT" = pmc_ew ] || [ "$WHAT" = all ]; then
  cd $REPO
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_ew -- python3 tools/pmc_ew.py > $OUT/fetch_ew.out 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_ew -- python3 tools/pmc_ew.py > $OUT/write_ew.out 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu_ew -- python3 tools/pmc_ew.py > $OUT/valu_ew.out 2>&1
fi
cd $REPO
for d in $OUT/pmc_*; do echo "== $d"; python3 tools/pmc_parse.py $d 2>/dev/null | cut -c1-400; done > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
