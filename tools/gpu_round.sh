#!/bin/bash
# One gpurun call = parity tests + the default bench + a rocprofv3 kernel-stats pass of the same
# bench command.  Usage (from the repo root, through gpurun):
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a [pytest-args...]'
# Everything lands under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
TAG=${1:-run}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c 'import __graft_entry__ as g; g.build()' > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -30 $OUT/build.log; exit 1; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  # one process per file: a GPU fault (abort) in one file must not hide the others' results
  : > $OUT/pytest.log; prc=0
  for f in ${TEST_FILES:-tests/test_*.py}; do
    timeout 900 python -m pytest $f -m gpu -q --durations=5 "$@" >> $OUT/pytest.log 2>&1
    r=$?; [ $r -ne 0 ] && [ $r -ne 5 ] && { prc=$r; echo "  $f rc=$r"; }
  done
  echo "pytest rc=$prc"; grep -E "^(FAILED|ERROR)|passed|failed|Fatal" $OUT/pytest.log | tail -40
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout 900 python bench.py ${BENCH_ARGS:---steps 20 --warmup 5} > $OUT/bench.json 2> $OUT/bench.err
  echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
fi
if [ "${SKIP_PROF:-0}" != "1" ]; then
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- \
      python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline ${PROF_ARGS:-} > $OUT/prof.log 2>&1 )
  echo "rocprof rc=$?"
  f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -40 $OUT/kernel_stats.csv | cut -c1-200
  t=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && [ -n "$t" ] && python tools/rocprof_roofline.py $OUT/kernel_stats.csv $t > $OUT/rocprof_roofline.txt 2>&1 \
    && cp $OUT/rocprof_roofline.json $OUT/rocprof_roofline_copy.json 2>/dev/null
  # keep the merge-back small: traces are large, the stats are what gets committed
  find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
fi
