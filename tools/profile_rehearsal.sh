#!/bin/bash
# DEV: kernel trace of the one-rank RCCL rehearsal (band of 1/8 frame): timeline of render and RCCL kernels in the timed region.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_reh
rm -rf $OUT && mkdir -p $OUT
cd $R
export BENCH_REHEARSE_DIST=1 BENCH_REHEARSE_BAND=${BAND:-8}
timeout -s KILL 120 rocprofv3 --kernel-trace -d $OUT/trace -o c3 -- python3 bench.py --steps 400 --warmup 10 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import sqlite3
con = sqlite3.connect("$OUT/trace/c3_results.db")
print("# one-rank rehearsal of the multi-GPU step (band of 1/${BAND:-8} of the C3 frame, RCCL loop-back gather per frame), rocprofv3 --kernel-trace")
for r in con.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size, count(*), avg(duration) from kernels where name like '%render_kernel%' or name like '%rccl%' group by name, grid_x"):
    print("kernel %-60s grid %7d threads, workgroup %4d, LDS %6d B, VGPRs(reported) %4d, SGPRs %4d, scratch %4d B, launches %5d, avg %8.1f us" % ((r[0][:60],) + tuple(r[1:8]) + (r[8] / 1e3,)))
rows = con.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
rows = [("render" if 'render_kernel' in r[0] else "rccl", r[1], r[2], r[3], r[4]) for r in rows if 'render_kernel' in r[0] or 'rccl' in r[0].lower()]
# the timed region: skip priming / warm-up at the front and the diagnostics at the back
mid = rows[int(len(rows) * 0.45):int(len(rows) * 0.75)]
win = mid[:40]
t0 = win[0][1]
print("# 40 consecutive kernels of the timed region (us)")
for n, s, e, q, st in win:
    print("%-7s start %8.1f  end %8.1f  dur %6.1f  stream %s" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, st))
rd = [(e - s) / 1e3 for n, s, e, q, st in mid if n == "render"]
rc = [(e - s) / 1e3 for n, s, e, q, st in mid if n == "rccl"]
rs = [s for n, s, e, q, st in mid if n == "render"]
print("# over %d frames: render kernel %.1f us avg, RCCL kernel %.1f us avg (min %.1f, max %.1f), render start to start %.1f us" % (len(rd), sum(rd) / len(rd), sum(rc) / len(rc), min(rc), max(rc), (rs[-1] - rs[0]) / 1e3 / (len(rs) - 1)))
PY
