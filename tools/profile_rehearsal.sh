#!/bin/bash
# DEV: kernel trace of the one-rank RCCL rehearsal (band of 1/8 frame): timeline of render and RCCL kernels.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_reh
rm -rf $OUT && mkdir -p $OUT
cd $R
export BENCH_REHEARSE_DIST=1 BENCH_REHEARSE_BAND=8
timeout -s KILL 120 rocprofv3 --kernel-trace -d $OUT/trace -o c3 -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import sqlite3
con = sqlite3.connect("$OUT/trace/c3_results.db")
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if 'kernel' in x.lower()]
print(t)
cur = con.execute("select * from kernels limit 1")
cols = [d[0] for d in cur.description]
print(cols)
rows = con.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
rows = [r for r in rows if 'render_kernel' in r[0] or 'rccl' in r[0]]
rows = rows[-60:]
t0 = rows[0][1]
for n, s, e, q, st in rows:
    print("%-8s start %9.1f us  dur %6.1f us  queue %s stream %s" % ("render" if 'render' in n else "rccl", (s - t0) / 1e3, (e - s) / 1e3, q, st))
PY
