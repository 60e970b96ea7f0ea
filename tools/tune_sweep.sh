#!/bin/bash
# DEV: sweep the RT_TUNE kernel variants on config C3 (run on the GPU box through gpurun).
mkdir -p gpurun_out
: > gpurun_out/tune.txt
for v in ${VARIANTS:-0 1 2 3 4 5 6 7 8 9}; do
  echo "== RT_TUNE=$v" >> gpurun_out/tune.txt
  RT_TUNE=$v timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline >> gpurun_out/tune.txt 2>&1 || exit 1
done
python - <<'PY'
import json
v=None
for line in open('gpurun_out/tune.txt'):
    if line.startswith('== '): v=line.strip()
    elif line.startswith('{'):
        d=json.loads(line); print(v, 'rays/s=%.3e'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], 'kern_ms=%.3f'%d['roofline']['kernel_ms'], 'TF=%.1f'%d['roofline']['achieved'], 'frac=%.3f'%d['roofline']['frac'])
PY
