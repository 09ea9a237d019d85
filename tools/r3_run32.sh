export TMPDIR=/tmp
o=gpurun_out/r3ad; mkdir -p $o
THIP_FUSE=0 timeout 1500 python -m pytest tests/test_gpu_frames.py -m gpu -x -q > $o/pytest_fuse0.log 2>&1; echo "fuse0 rc=$?"; tail -2 $o/pytest_fuse0.log
THIP_SKIP_STATIC=2 timeout 500 python tests/soak_parity.py 17 300 > $o/soak2.log 2>&1; echo "soak skip_static=2 rc=$?"; tail -1 $o/soak2.log
THIP_FUSE=0 timeout 500 python tests/soak_parity.py 19 300 > $o/soak3.log 2>&1; echo "soak fuse0 rc=$?"; tail -1 $o/soak3.log
THIP_FE_DEVICE_LISTS=1 timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_thirdparty_decoder.py -m gpu -x -q > $o/pytest_lists.log 2>&1; echo "frontend under lists rc=$?"; tail -2 $o/pytest_lists.log
