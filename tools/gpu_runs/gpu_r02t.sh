mkdir -p gpurun_out/r02t
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_models_gpu.py::test_graphed_detector_equals_eager -x -q 2>&1 | grep -E "assert|Error|passed|failed|it, k|\(it" | head -12) > gpurun_out/r02t/alone.log
(timeout 600 python -m pytest tests/test_full_size_gpu.py tests/test_models_gpu.py::test_graphed_detector_equals_eager -x -q 2>&1 | grep -E "assert|Error|passed|failed|\(it" | head -12) > gpurun_out/r02t/after_full.log
(TF_NO_MHA=1 timeout 300 python -m pytest tests/test_models_gpu.py::test_graphed_detector_equals_eager -x -q 2>&1 | grep -E "assert|passed|failed" | head -5) > gpurun_out/r02t/nomha.log
(TF_SPLIT_LINEAR=0 timeout 300 python -m pytest tests/test_models_gpu.py::test_graphed_detector_equals_eager -x -q 2>&1 | grep -E "assert|passed|failed" | head -5) > gpurun_out/r02t/nosplit.log
