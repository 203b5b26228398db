#!/bin/bash
# Round-6 profile set (run on the GPU box; outputs in gpurun_out/prof6/, copy
# what is to be kept into profiles/):
#  * bench: rocprofv3 --kernel-trace of the DEFAULT headline command (5 + 20
#    steps, no configs / cpu baseline); the summary is restricted to the TIMED
#    WINDOW (the two tce_marker launches bench.py puts around its K steps), per
#    kernel x workgroup count; the traced run's own JSON line is kept beside it
#    (r06_bench_line_traced.json), so  sum(dominant kernel) / steps  can be held
#    against that run's ms_per_step.
#  * the `configs` entries named on the command line: whole-run summaries.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CONFIGS=${@:-"bench C4_bbrl_shard"}
for c in $CONFIGS; do
  rm -rf /tmp/tr_$c
  if [ $c = bench ]; then
    timeout -k 10 600 rocprofv3 --kernel-trace -d /tmp/tr_$c -o p -- python3 $R/bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $OUT/r06_bench_line_traced.json 2> $OUT/bench.err || { echo "trace bench failed"; tail -5 $OUT/bench.err; exit 1; }
    db=$(find /tmp/tr_$c -name "*.db" | head -1)
    python3 $R/scripts/rocpd_stats.py $db $OUT/r06_bench_kernel_stats_by_grid.csv --by-grid --between-markers 1 2 | tee $OUT/bench_window.txt
    python3 $R/scripts/rocpd_stats.py $db $OUT/r06_bench_kernel_stats.csv --between-markers 1 2
  else
    timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/tr_$c -o p -- python3 $R/scripts/run_config.py $c 3 2 nofloor > $OUT/$c.log 2>&1 || { echo "trace $c failed"; tail -5 $OUT/$c.log; exit 1; }
    db=$(find /tmp/tr_$c -name "*.db" | head -1)
    python3 $R/scripts/rocpd_stats.py $db $OUT/r06_${c}_kernel_stats.csv
    python3 $R/scripts/rocpd_stats.py $db $OUT/r06_${c}_kernel_stats_by_grid.csv --by-grid || true
    python3 $R/scripts/rocpd_gaps.py $db > $OUT/r06_${c}_gaps.txt 2>&1 || true
    if [ $c = C4_bbrl_shard ]; then
      # the policy epoch's dependent chain (row kernel -> slab reduction -> finish) and,
      # beside it on the second stream, the critic's (row kernel -> reduction)
      python3 $R/scripts/rocpd_chain.py $db "smlp_epoch_kernel<32, 4, 2" smlp_reduce_kernel bb_diag_finish_kernel --skip-frac 0.5 | tee $OUT/r06_C4_bbrl_shard_chain.json
      python3 $R/scripts/rocpd_chain.py $db "smlp_epoch_kernel<32, 1, 1" smlp_reduce_kernel --skip-frac 0.5 | tee $OUT/r06_C4_bbrl_shard_chain_critic.json
    fi
  fi
  rm -rf /tmp/tr_$c
  echo "trace $c done"
done
ls -la $OUT | head -30
