#!/bin/bash
# Profile the C3 bench on the GPU box: kernel trace + stats, then PMC passes (each in its own run).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o c3 -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/pmc_$name -o c3 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_pmc_$name.log 2>&1; }
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
pass f32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT
pass hbmr TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
pass hbmw TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
find $OUT -name "*.csv" | head -50
