set -e
for rep in 1 2; do
for plan in 0; do
for b in 8 4 2; do
RT_SMALL_PLAN=$plan BENCH_REHEARSE_DIST=1 BENCH_REHEARSE_BAND=$b timeout -k 10 200 python bench.py --steps 2000 --no-cpu-baseline > gpurun_out/rp.json 2> gpurun_out/rp.err
python - <<PY
import json
d=json.load(open("gpurun_out/rp.json")); print("plan=$plan band=1/$b us/step=%.1f render-only=%.1f gather-only=%.1f" % (d["ms_per_step"]*1e3, d["diagnostics"]["render_only_us_per_step"], d["diagnostics"]["gather_only_us_per_step"]))
PY
done; done; done
