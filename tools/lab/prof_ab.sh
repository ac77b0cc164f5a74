#!/bin/bash
# tools/lab/prof_ab.sh <tag> <base.so> <cand.so> [<cand2.so> ...]: rocprofv3 kernel traces of the same
# bench command under several builds of the C-ABI library in ONE lease (box-to-box variance is
# 1.5 - 3 %), then tools/lab/trace_ab.py base vs every candidate: durations per LAYER of the captured step.
#   PATTERN=<substring of the kernel names to list> (default conv_gemm_glds)   BENCH_ARGS="--config c4"
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
for L in "$@"; do
  n=$(basename $L .so)
  ( cd /tmp && SEGMENTRON_HIP_LIB=$ROOT/$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -- \
      python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs --no-parity ${BENCH_ARGS:-} > $OUT/prof_$n.log 2>&1 )
  f=$(find $OUT/prof_$n -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$n.csv
  t=$(find $OUT/prof_$n -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && cp $t /tmp/trace_$n.csv
  rm -rf $OUT/prof_$n
done
A=$(basename $1 .so); shift
for L in "$@"; do
  n=$(basename $L .so)
  echo "==== $A -> $n"
  python tools/lab/trace_ab.py /tmp/trace_$A.csv /tmp/trace_$n.csv "${PATTERN:-conv_gemm_glds}" | tee $OUT/trace_ab_$n.txt
done
