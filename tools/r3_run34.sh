export TMPDIR=/tmp
for t in 2 4 8; do for k in dense typical; do for env in "THIP_FE_DEVICE_LISTS=1" ""; do
v=$(env $env timeout 300 python bench.py --mode e2e --e2e-size 720p --packets $k --threads $t --no-native --loops 6 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])"); echo "$t threads $k [$env]: $v"; done; done; done
