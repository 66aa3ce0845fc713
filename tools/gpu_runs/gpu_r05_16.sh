cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r16
timeout 600 python -m pytest tests/test_fused_gpu.py -q -k "mha or self_attention" 2>&1 | tail -8 > gpurun_out/r16/mha_tests.txt
timeout 300 python tools/bench_mha.py > gpurun_out/r16/bench_mha.txt 2>&1
timeout 300 python tools/experiments/conv3_tiles.py > gpurun_out/r16/conv3_tiles.txt 2>&1
cat gpurun_out/r16/*.txt
