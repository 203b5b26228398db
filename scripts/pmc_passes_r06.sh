#!/bin/bash
# round 6: three rocprofv3 --pmc passes (separate runs, program directly after
# `--`) over scripts/pmc_kernels.py (the roofline kernels, cold caches):
#   FETCH_SIZE, WRITE_SIZE                     -> HBM bytes per launch
#   SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   -> matrix-pipe utilisation
# results under gpurun_out/pmc6/<pass>/; summarise here with
#   python scripts/pmc_summarize.py gpurun_out/pmc6/FETCH_SIZE/.../p_counter_collection.csv ... r06
#   python scripts/pmc_mfma_summarize.py gpurun_out/pmc6/MFMA r06
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc6/$c -o p -- python3 $R/scripts/pmc_kernels.py > $R/gpurun_out/pmc6_$c.log 2>&1 || echo "pass $c failed"
  echo "pass $c done"
done
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc6/MFMA -o p -- python3 $R/scripts/pmc_kernels.py > $R/gpurun_out/pmc6_MFMA.log 2>&1 || echo "pass MFMA failed"
echo "pass MFMA done"
find $R/gpurun_out/pmc6 -name "*counter_collection.csv" | head
