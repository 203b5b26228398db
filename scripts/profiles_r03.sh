#!/bin/bash
# Round-3 profile set (run on the GPU box; summaries land in gpurun_out/prof3/,
# copy the ones to keep into profiles/):
#   kernel traces (rocprofv3 --kernel-trace, rocpd database summarised by
#   scripts/rocpd_stats.py) of the bench headline run and of every `configs`
#   entry; then the FETCH_SIZE / WRITE_SIZE passes over scripts/pmc_kernels.py.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
trace() {   # tag, script args...
  local tag=$1; shift
  rm -rf /tmp/tr_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/tr_$tag -o p -- python3 "$@" > $OUT/$tag.log 2>&1 || { echo "trace $tag failed"; return 1; }
  python3 $R/scripts/rocpd_stats.py $(find /tmp/tr_$tag -name "*.db" | head -1) $OUT/r03_${tag}_kernel_stats.csv
  rm -rf /tmp/tr_$tag
  echo "trace $tag done"
}
trace bench $R/bench.py --no-cpu-baseline --no-configs || exit 1
for c in C3_box_push_f32 C3_box_push_f64 C4_bbrl_shard C5_table_tennis_nb3_shard C5_table_tennis_nb8_shard; do
  trace $c $R/scripts/run_config.py $c 3 2 || exit 1
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/scripts/pmc_kernels.py > $OUT/pmc_$c.log 2>&1 || { echo "pass $c failed"; exit 1; }
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $OUT/pmc_$c.csv
  echo "pass $c done"
done
# (summarise here: python scripts/pmc_summarize.py gpurun_out/prof3/pmc_FETCH_SIZE.csv gpurun_out/prof3/pmc_WRITE_SIZE.csv r03)
ls -la $OUT | head -30
