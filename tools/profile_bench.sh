#!/bin/bash
# Profile one bench.py command on the GPU box: kernel trace + stats, then PMC passes (each in its own run, no other tracing).
# Usage: tools/profile_bench.sh <tag> [bench.py args...]   e.g.  tools/profile_bench.sh c3_depth2 --config C3 --depth 2
# Then on the CPU side: python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/r03_<tag>
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $R
echo "python3 bench.py $* --no-cpu-baseline (trace: --steps 50 --warmup 5; pmc: --steps 10 --warmup 2)" > $OUT/command.txt
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $OUT/trace -o c3 -- python3 bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
pass() { name=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" -d $OUT/pmc_$name -o c3 -- python3 bench.py $ARGS --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_pmc_$name.json 2> $OUT/bench_pmc_$name.err; }
ARGS="$*"
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
pass f32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT
pass hbmr TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
pass hbmw TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
cat $OUT/bench_trace.json
