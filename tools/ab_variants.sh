#!/bin/bash
# DEV: A/B of library variants built into build_variants/*.so (same box, one gpurun call): tools/ab_variants.sh "<cmd>" v1 v2 ...
cmd=$1; shift
cp raytracing_amd/librt_mi355x.so /tmp/keep.so
for rep in 1 2; do for v in "$@"; do
  cp build_variants/$v.so raytracing_amd/librt_mi355x.so
  echo "== $v"; timeout -k 10 120 $cmd || { cp /tmp/keep.so raytracing_amd/librt_mi355x.so; exit 1; }
done; done
cp /tmp/keep.so raytracing_amd/librt_mi355x.so
