#!/bin/bash
# Round-4 profile set (run on the GPU box; summaries land in gpurun_out/prof4/,
# copy the ones to keep into profiles/): kernel traces (rocprofv3 --kernel-trace,
# rocpd database summarised by scripts/rocpd_stats.py, per kernel and per kernel x
# workgroup count) of the bench headline run and of the `configs` entries named
# on the command line (default: all single-GPU shards).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
trace() {   # tag, script args...
  local tag=$1; shift
  rm -rf /tmp/tr_$tag
  timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/tr_$tag -o p -- python3 "$@" > $OUT/$tag.log 2>&1 || { echo "trace $tag failed"; tail -5 $OUT/$tag.log; return 1; }
  local db=$(find /tmp/tr_$tag -name "*.db" | head -1)
  python3 $R/scripts/rocpd_stats.py $db $OUT/r04_${tag}_kernel_stats.csv
  python3 $R/scripts/rocpd_stats.py $db $OUT/r04_${tag}_kernel_stats_by_grid.csv --by-grid || true
  rm -rf /tmp/tr_$tag
  echo "trace $tag done"
}
CONFIGS=${@:-"bench C3_box_push_f32 C3_box_push_f64 C4_bbrl_shard C5_table_tennis_nb3_shard C5_table_tennis_nb8_shard"}
for c in $CONFIGS; do
  if [ $c = bench ]; then
    trace bench $R/bench.py --no-cpu-baseline --no-configs || exit 1
  else
    trace $c $R/scripts/run_config.py $c 3 2 || exit 1
  fi
done
ls -la $OUT | head -30
