cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r23
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_dropin_compiled_gpu.py -q -k "capturable" 2>&1 | tail -1; done > gpurun_out/r23/captured_backward_6x.txt
timeout 300 python tools/experiments/conv3_tiles.py --ti 4 > gpurun_out/r23/conv3_tiles_128_rows.txt 2>&1
cat gpurun_out/r23/*.txt
