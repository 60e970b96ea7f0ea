#!/bin/bash
# DEV: kernel trace of the one-rank RCCL rehearsal (band of 1/8 frame): how long do the RCCL kernels run next to the render kernels?
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_reh
rm -rf $OUT && mkdir -p $OUT
cd $R
export BENCH_REHEARSE_DIST=1 BENCH_REHEARSE_BAND=8
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o c3 -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import sqlite3
con = sqlite3.connect("$OUT/trace/c3_results.db")
for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 6"):
    print("%-90s calls %5d avg %.1f us  %.1f %%" % (name[:90], calls, avg/1000 if avg > 1e4 else avg, pct))
PY
