#!/bin/bash
# does SQ_INSTS_VALU_FMA_F32 count a v_pk_fma_f32 as one or as two instructions?  Runs the valu_rate micro-benchmark
# (16 FMA-class instructions per loop iteration in every mode) under one rocprofv3 --pmc pass and prints the counter per kernel.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_pkcal
rm -rf $OUT && mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o $OUT/valu_rate $R/tools/microbench/valu_rate.hip
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU SQ_INSTS_VALU_MUL_F32 -d $OUT/pmc -o cal -- $OUT/valu_rate > $OUT/run.log 2>&1
python3 - <<PY
import sqlite3, glob
db = glob.glob("$OUT/pmc/*results.db")[0]
c = sqlite3.connect(db)
rows = c.execute("select kernel_name, counter_name, avg(value), max(value) from counters_collection group by kernel_name, counter_name").fetchall()
by = {}
for k, n, a, m in rows: by.setdefault(k, {})[n] = m
for k, v in by.items():
    print(k[:70].ljust(70), {n: "%.4g" % x for n, x in v.items()})
PY
