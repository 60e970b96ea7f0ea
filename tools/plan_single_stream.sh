set -e
for rep in 1 2; do
for plan in 0 1 3 4; do
echo "plan=$plan"
RT_SMALL_PLAN=$plan HS=135,270,540,1080 timeout -k 10 120 python tools/time_size_b2b.py 2>&1 | cut -c1-60
done; done
