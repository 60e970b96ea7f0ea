#!/bin/bash
# DEV: sweep the tile-section fractions (RT_TUNE4 / RT_TUNE2, permille of rows in 16- and 8-row tiles) on config C3 through gpurun.
mkdir -p gpurun_out
: > gpurun_out/tune.txt
for v in ${VARIANTS:-"850 100"}; do
  set -- $v
  echo "== RT_TUNE4=$1 RT_TUNE2=$2" >> gpurun_out/tune.txt
  RT_TUNE4=$1 RT_TUNE2=$2 timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline >> gpurun_out/tune.txt 2>&1 || exit 1
done
python - <<'PY'
import json
v=None
for line in open('gpurun_out/tune.txt'):
    if line.startswith('== '): v=line.strip()
    elif line.startswith('{'):
        d=json.loads(line); print(v, 'rays/s=%.3e'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], 'kern_ms=%.3f'%d['roofline']['kernel_ms'], 'TF=%.1f'%d['roofline']['achieved'], 'frac=%.3f'%d['roofline']['frac'])
PY
