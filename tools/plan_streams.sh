set -e
for rep in 1 2; do
for plan in 0 1 3; do
echo "plan=$plan"
RT_SMALL_PLAN=$plan timeout -k 10 120 python tools/two_stream.py 2>&1 | grep rows=
done; done
