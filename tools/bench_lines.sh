#!/bin/bash
# One gpurun call: the plain (unprofiled) bench.py line of every command the round reports -> gpurun_out/r03_bench_<tag>.json
set -e
cd $GRAFT_REPO_ROOT
one() { tag=$1; shift; timeout -k 10 300 python bench.py "$@" > gpurun_out/r03_bench_$tag.json 2> gpurun_out/bench_$tag.err; python - <<PY
import json
d = json.loads(open("gpurun_out/r03_bench_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
}
one c3
one c3_depth2 --config C3 --depth 2
one c4_primary --config C4
one c5_primary --config C5
one c5_shadows --config C5 --shadows
one c3_cull --config C3 --cull
one c3_depth2_cull --config C3 --depth 2 --cull
one c4_cull --config C4 --cull
one c5_cull --config C5 --cull
one c5_shadows_cull --config C5 --shadows --cull
