#!/bin/bash
# tools/lab/bench_ab.sh <tag> <base.so> <cand.so> <config> [config ...]: bench.py of every config
# under two builds of the C-ABI library, alternating, twice each, in ONE lease
TAG=$1; A=$2; B=$3; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
for c in "$@"; do
  for i in 1 2; do
    for L in $A $B; do
      r=$(SEGMENTRON_HIP_LIB=$PWD/$L python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms %.1f %s' % (d['ms_per_step'], d['value'], d['unit']))")
      echo "$c $(basename $L) $r" | tee -a $OUT/bench_ab.txt
    done
  done
done
