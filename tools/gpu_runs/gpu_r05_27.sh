cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r27
timeout 600 python -m pytest tests/test_models_gpu.py -q -k "prepare or pipelined or graphed" 2>&1 | tail -15 > gpurun_out/r27/prepare_tests.txt
cat gpurun_out/r27/prepare_tests.txt
