cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tl -o p -- python3 $R/bench.py --no-cpu-baseline --no-configs --steps 3 --warmup 3 > /tmp/tl.log 2>&1 || { tail -5 /tmp/tl.log; exit 1; }
db=$(find /tmp/tl -name "*.db" | head -1)
python3 - "$db" <<'P'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in con.execute("select name from sqlite_master where type='table' or type='view'")]
kd=[t for t in tabs if 'kernel_dispatch' in t and 'rocpd' in t][0]
ks=[t for t in tabs if 'kernel_symbol' in t or 'info_kernel_symbol' in t]
print(kd, ks[:3], file=sys.stderr)
cols=[r[1] for r in con.execute("pragma table_info(%s)"%kd)]
print(cols, file=sys.stderr)
P
python3 $R/scripts/rocpd_stats.py $db /tmp/x.csv >/dev/null 2>&1
python3 - "$db" <<'P'
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in con.execute("select name from sqlite_master")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=con.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.stream_id from %s d join %s s on d.kernel_id = s.id order by d.start"%(kd,sym)).fetchall()
# find the policy epochs of the LAST step: kernels named policy_tail_kernel
idx=[i for i,r in enumerate(rows) if 'policy_tail_kernel' in r[0]]
a,b=idx[-12],idx[-11]
ep=rows[a+1:b+1]
t0=ep[0][1]
for n,s,e,g,w,st in ep:
    if 'mlp_critic_bwd_kernel<1, 10, false>' in n or 'mlp_finish' in n and g>10000: continue
    nm=re.sub(r'\(anonymous namespace\)::','',n)[:70]
    print("%8.1f %7.1f  s%-3d g%-6d %s"%((s-t0)/1e3,(e-s)/1e3,st,g//max(w,1),nm))
print("span %.1f us"%((ep[-1][2]-t0)/1e3))
P
