#!/bin/bash
# SQ counter passes over the critic kernel; results under gpurun_out/pmc_mlp/<pass>/
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d /root/repo/gpurun_out/pmc_mlp/p$i -o p -- python3 /root/repo/scripts/pmc_mlp.py > /root/repo/gpurun_out/pmc_mlp_p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
