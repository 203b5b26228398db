#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bb
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_bb
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr_bb -o p -- python3 $R/scripts/time_bbrl_docs.py 4096 hand 5 $1 > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
db=$(find /tmp/tr_bb -name "*.db" | head -1)
python3 $R/scripts/rocpd_stats.py $db $OUT/r04_bbrl_docs${1:+_$1}_kernel_stats.csv
tail -3 $OUT/trace.log
head -40 $OUT/r04_bbrl_docs${1:+_$1}_kernel_stats.csv | cut -c1-160
