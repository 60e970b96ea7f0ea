set -e
for rep in 1 2; do
for c in 1 3; do
for b in 8 4; do
BENCH_COMMS=$c BENCH_REHEARSE_DIST=1 BENCH_REHEARSE_BAND=$b timeout -k 10 200 python bench.py --steps 2000 --no-cpu-baseline > gpurun_out/reh_c${c}_b${b}.json 2> gpurun_out/reh_c${c}_b${b}.err
python - <<PY
import json
d=json.load(open("gpurun_out/reh_c${c}_b${b}.json")); print("comms=$c band=1/$b us/step=%.1f" % (d["ms_per_step"]*1e3), d["config"]["parallelism"])
PY
done; done; done
