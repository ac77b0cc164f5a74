#!/bin/bash
# rocprofv3 PMC passes over a stand-alone lab binary (one counter set per pass, own timeout each);
# prints per-kernel means of every counter.
#   bash tools/lab/pmc_lab.sh <tag> <kernel-name substring> -- <binary> [args...]
TAG=$1; PAT=$2; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
CMD="$PWD/$1"; shift
case "$CMD" in *.py) CMD="python $CMD";; esac
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- $CMD "$@" > $OUT/p$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:70], r.get("Grid_Size", ""))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    if pat not in k[0]: continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
