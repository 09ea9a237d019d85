export TMPDIR=/tmp
o=gpurun_out/r3q; mkdir -p $o
sel='(sequence or enqueue or batched or grey or dup or lane_shared or static_background or four_concurrent or beyond_4k or frame_calls or config3 or gop_parallel) and not elision'
timeout 1500 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "$sel" > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $o/pytest.log
bash tools/ab.sh base=tools/_build/libtheora_hip_base.so bot2=theora_amd/libtheora_hip.so 2>&1 | tee $o/ab.txt
