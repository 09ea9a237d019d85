export TMPDIR=/tmp
o=gpurun_out/r3aa; mkdir -p $o
sel='(sequence or enqueue or batched or grey or dup or lane_shared or static_background or four_concurrent or beyond_4k or frame_calls) and not elision'
timeout 1500 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "$sel" > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.log
bash tools/ab.sh base=tools/_build/libtheora_hip_base.so new=theora_amd/libtheora_hip.so 2>&1 | tee $o/ab.txt
