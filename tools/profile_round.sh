#!/bin/bash
# One gpurun call: the whole -m gpu suite, then the round's profiles (kernel trace + PMC passes) and plain bench lines of every config.
# Usage on the GPU box: bash tools/profile_round.sh     (summaries: python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/r03_<tag>)
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3_suite.log 2>&1 || { tail -20 gpurun_out/r3_suite.log; exit 1; }
tail -2 gpurun_out/r3_suite.log
# the rocpd databases are far too large to travel back (gpurun_out is capped at 64 MiB): summarise on the box, keep the summaries and
# bench.py's JSON lines, drop the databases
one() {
    tag=$1; shift
    bash tools/profile_bench.sh $tag "$@" > /dev/null 2>&1
    python tools/summarize_prof.py gpurun_out/prof_$tag gpurun_out/r03_$tag > /dev/null
    cp gpurun_out/prof_$tag/bench_trace.json gpurun_out/r03_${tag}_bench_under_rocprof.json
    rm -rf gpurun_out/prof_$tag
    echo "$tag profiled"
}
one c3 --config C3
one c3_depth2 --config C3 --depth 2
one c5_shadows --config C5 --shadows
one c5 --config C5
one c4 --config C4
one c3_cull --config C3 --cull
one c3_depth2_cull --config C3 --depth 2 --cull
one c5_shadows_cull --config C5 --shadows --cull
one c4_cull --config C4 --cull
one c5_cull --config C5 --cull
