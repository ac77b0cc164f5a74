#!/bin/bash
# tools/lab/prof_ab.sh <tag> <libA.so> <libB.so> [bench args]: rocprofv3 kernel stats of the same bench
# command under two builds of the C-ABI library, in ONE lease (box-to-box variance is 1.5 - 3 %)
TAG=$1; A=$2; B=$3; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
for L in $A $B; do
  n=$(basename $L .so)
  ( cd /tmp && SEGMENTRON_HIP_LIB=$ROOT/$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -- \
      python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs --no-parity "$@" > $OUT/prof_$n.log 2>&1 )
  f=$(find $OUT/prof_$n -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$n.csv
  t=$(find $OUT/prof_$n -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && cp $t /tmp/trace_$n.csv
  rm -rf $OUT/prof_$n
  tail -1 $OUT/prof_$n.log | cut -c1-200
done
python tools/lab/trace_ab.py /tmp/trace_$(basename $A .so).csv /tmp/trace_$(basename $B .so).csv ${PATTERN:-conv_gemm_glds} | tee $OUT/trace_ab.txt
