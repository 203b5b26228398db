#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the roofline kernels (separate runs, as
# the counters do not fit one pass); summaries -> gpurun_out/pmc/{fetch,write}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$c -o p -- python3 $GRAFT_REPO_ROOT/scripts/pmc_kernels.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1 || echo "pass $c failed"
  echo "pass $c done"
done
