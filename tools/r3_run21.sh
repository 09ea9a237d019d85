export TMPDIR=/tmp
o=gpurun_out/r3t; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu.log
timeout 700 python tests/soak_parity.py 11 500 > $o/soak.log 2>&1; echo "soak rc=$?"; tail -2 $o/soak.log
