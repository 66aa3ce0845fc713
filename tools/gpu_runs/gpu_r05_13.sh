mkdir -p gpurun_out/r05_13
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_models_gpu.py -m gpu -q -x 2>&1 | grep -v Warning | tail -70 | tee gpurun_out/r05_13/pytest.txt
