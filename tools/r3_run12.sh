export TMPDIR=/tmp
o=gpurun_out/r3j; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_frames.py -m gpu -x -q > $o/pytest_frames.log 2>&1; echo "frames rc=$?"; tail -3 $o/pytest_frames.log
bash tools/ab.sh units=theora_amd/libtheora_hip.so base=tools/_build/libtheora_hip_base.so 2>&1 | tee $o/ab.txt
python tools/lf_trace.py --content smooth 2>&1 | grep -v amdgpu.ids | head -14 > $o/lf_trace_smooth.txt; cat $o/lf_trace_smooth.txt | cut -c1-120
