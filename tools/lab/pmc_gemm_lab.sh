#!/bin/bash
# rocprofv3 PMC passes over the stand-alone GEMM lab (one counter set per pass, own timeout each).
#   bash tools/lab/pmc_gemm_lab.sh <tag> <shape idx> <which: 0 px256 | 1 glds>
TAG=${1:-pmc}; SH=${2:-0}; W=${3:-1}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
BIN=$PWD/tools/lab/gemm_lab
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  LAB_SHAPES=$SH LAB_NOTEST=1 LAB_WHICH=$W timeout 60 rocprofv3 --pmc $set --kernel-trace \
     --output-format csv -d $OUT/p$i -- $BIN 6 > $OUT/p$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "conv_gemm" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
