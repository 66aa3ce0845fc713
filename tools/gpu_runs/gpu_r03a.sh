mkdir -p gpurun_out/r03a
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_models_gpu.py tests/test_full_size_gpu.py -x -q -k "graph or full_size" 2>&1 | tail -4) > gpurun_out/r03a/pytest.log
