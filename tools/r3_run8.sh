export TMPDIR=/tmp
o=gpurun_out/r3f; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_slots.py -m gpu -x -q -k "dc_unpredict" > $o/pytest_slots.log 2>&1; echo "slots rc=$?"; tail -3 $o/pytest_slots.log
timeout 1200 python -m pytest tests/test_gpu_frames.py tests/test_gpu_frontend.py -m gpu -x -q -k "dc_unprediction or (4k and device_dc)" > $o/pytest_dc.log 2>&1; echo "dc rc=$?"; tail -3 $o/pytest_dc.log
timeout 600 python tools/dc_wavefront_time.py 2>&1 | grep -v amdgpu.ids | tee $o/dc_wavefront_time.txt
for th in 1 16; do for mode in "" THIP_FE_DEVICE_DC=1 THIP_FE_DEVICE_LISTS=1; do
  env $mode python bench.py --mode e2e --e2e-size 720p --packets dense --threads $th --loops 4 --no-native 2>/dev/null | grep '^{' | head -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('e2e 720p dense threads $th [$mode]:', d['value'], 'fps')" | tee -a $o/e2e_modes.txt
done; done
