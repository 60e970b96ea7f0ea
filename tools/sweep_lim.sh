set -e
for rep in 1 2; do
for lim in 30 10 8; do
for b in 8 4; do
RT_LIM=$lim BENCH_REHEARSE_DIST=1 BENCH_REHEARSE_BAND=$b timeout -k 10 100 python bench.py --steps 1500 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lim=$lim band=1/$b pipeline us/step=%.1f' % (d['ms_per_step']*1e3))"
done
done; done
