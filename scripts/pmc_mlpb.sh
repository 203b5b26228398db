#!/bin/bash
# counter passes over the three-part bf16 critic kernel (csrc/mlpb.hip); results under
# gpurun_out/pmc_mlpb/<pass>/ (summarised into profiles/r05_pmc_mlpb.json by hand)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_mlpb/p$i -o p -- python3 $GRAFT_REPO_ROOT/scripts/pmc_mlp.py bf16x3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_mlpb_p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
# HBM bytes (separate passes, as MI355X_MICROARCH.md prescribes): FETCH_SIZE, WRITE_SIZE
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_mlpb/$c -o p -- python3 $GRAFT_REPO_ROOT/scripts/pmc_mlp.py bf16x3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_mlpb_$c.log 2>&1 || echo "pass $c failed"
  echo "pass $c done"
done
