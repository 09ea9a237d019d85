export TMPDIR=/tmp
o=gpurun_out/r3ae; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_frontend.py -m gpu -x -q -k "token_lists or 4k" > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.log
for i in 1 2 3; do for env in "THIP_FE_DEVICE_LISTS=1" ""; do
v=$(env $env timeout 300 python bench.py --mode e2e --e2e-size 720p --packets dense --no-native --loops 8 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])"); echo "1 thread dense [$env]: $v"; done; done
for env in "THIP_FE_DEVICE_LISTS=1" ""; do
v=$(env $env timeout 300 python bench.py --mode e2e --e2e-size 720p --packets typical --no-native --loops 8 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])"); echo "1 thread typical [$env]: $v"
v=$(env $env timeout 300 python bench.py --mode e2e --e2e-size 720p --packets dense --threads 16 --no-native --loops 6 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])"); echo "16 threads dense [$env]: $v"; done
THIP_FE_DEVICE_LISTS=1 THIP_FE_PROF=1 timeout 300 python bench.py --mode e2e --e2e-size 720p --packets dense --no-native --loops 8 2>&1 | grep -v "^{\|amdgpu.ids" | head -10
