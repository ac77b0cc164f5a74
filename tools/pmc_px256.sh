#!/bin/bash
# rocprofv3 PMC passes for conv_gemm_px256_kernel on 728->728 @2x65x129 (one counter set per pass,
# every pass under its own timeout: a rejected counter set makes rocprofv3 hang on abort).
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_px256
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PROBE_ONLY=0 PROBE_ITERS=6 SEG_GEMM_L2_WARM=0 timeout 90 rocprofv3 --pmc $set --kernel-trace \
     --output-format csv -d $OUT/p$i -- python /root/repo/tools/gemm_probe.py > $OUT/p$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
