set -e
for rep in 1 2; do
for cfg in "850 100" "960 30" "970 25" "980 20"; do
set -- $cfg
for c in C3 C5 C4; do
st=400; [ $c = C4 ] && st=30; [ $c = C5 ] && st=150
RT_S4=$1 RT_S2=$2 timeout -k 10 100 python bench.py --config $c --steps $st --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('s4=$1 s2=$2 $c ms/step=%.4f kernel_ms=%.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done
RT_S4=$1 RT_S2=$2 BENCH_REHEARSE_DIST=1 BENCH_REHEARSE_BAND=2 timeout -k 10 100 python bench.py --steps 1000 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('s4=$1 s2=$2 half-band pipeline us/step=%.1f' % (d['ms_per_step']*1e3))"
done; done
