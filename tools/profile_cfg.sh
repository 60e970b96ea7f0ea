#!/bin/bash
# Profile one tools/time_config.py run: kernel trace + divergence / VALU counters.  Usage: tools/profile_cfg.sh <tag> <time_config args...>
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o c3 -- python3 tools/time_config.py "$@" --frames 2 > $OUT/trace.log 2>&1
pass() { name=$1; shift; timeout -s KILL 300 rocprofv3 --pmc "$@" -d $OUT/pmc_$name -o c3 -- python3 tools/time_config.py $ARGS --frames 1 > $OUT/pmc_$name.log 2>&1; }
ARGS="$@"
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
pass f32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT
tail -1 $OUT/trace.log
