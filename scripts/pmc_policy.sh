#!/bin/bash
# SQ counter passes over the policy epochs alone (scripts/prof_policy.py);
# results under gpurun_out/pmc_pol/p<i>/ (summarised by scripts/pmc_policy_summarize.py)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_pol/p$i -o p -- python3 $GRAFT_REPO_ROOT/scripts/prof_policy.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_pol_p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
