set -e
for rep in 1 2; do
for sl in 3 4; do
for b in 8 4 2; do
BENCH_SLOTS=$sl BENCH_REHEARSE_DIST=1 BENCH_REHEARSE_BAND=$b timeout -k 10 150 python bench.py --steps 2000 --no-cpu-baseline > gpurun_out/rp.json 2> gpurun_out/rp.err
python - <<PY
import json
d=json.load(open("gpurun_out/rp.json")); print("slots=$sl band=1/$b us/step=%.1f render-only=%.1f gather-only=%.1f %s" % (d["ms_per_step"]*1e3, d["diagnostics"]["render_only_us_per_step"], d["diagnostics"]["gather_only_us_per_step"], d["config"]["parallelism"]))
PY
done; done; done
BENCH_GATHER=torch BENCH_REHEARSE_DIST=1 BENCH_REHEARSE_BAND=8 timeout -k 10 150 python bench.py --steps 2000 --no-cpu-baseline 2> gpurun_out/rp.err | cut -c1-200
grep -h "concurrent\|probe" gpurun_out/rp.err || true
