#!/bin/bash
# LDS counter passes over the split-f16 critic kernel; results under gpurun_out/pmc_mlp16/<pass>/
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d /root/repo/gpurun_out/pmc_mlp16/p$i -o p -- python3 /root/repo/scripts/pmc_mlp.py f16x2 > /root/repo/gpurun_out/pmc_mlp16_p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
