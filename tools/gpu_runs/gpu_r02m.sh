mkdir -p gpurun_out/r02m
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_fused_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/r02m/pytest_fused.log
(timeout 120 python tools/bench_mha.py 2>&1 | tail -3) > gpurun_out/r02m/mha.log
(timeout 600 python bench.py --no-cpu-baseline 2>/dev/null) > gpurun_out/r02m/bench.json
