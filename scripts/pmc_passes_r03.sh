#!/bin/bash
# round 3: FETCH_SIZE / WRITE_SIZE passes (separate runs) over
#   scripts/pmc_kernels.py      (C2 / C3 roofline kernels, cold caches)  -> gpurun_out/pmc3/main
#   scripts/pmc_kernels_big.py  (GAE / trajectory at 8 x the envs)       -> gpurun_out/pmc3/big
#   scripts/pmc_bbrl.py         (black-box agent's row kernels)          -> gpurun_out/pmc3/bbrl
cd /tmp && export TMPDIR=/tmp
for s in main:pmc_kernels.py big:pmc_kernels_big.py bbrl:pmc_bbrl.py; do
  tag=${s%%:*}; script=${s##*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3/$tag/$c -o p -- python3 $GRAFT_REPO_ROOT/scripts/$script > $GRAFT_REPO_ROOT/gpurun_out/pmc3/${tag}_$c.log 2>&1 || echo "pass $tag $c failed"
    echo "pass $tag $c done"
  done
done
