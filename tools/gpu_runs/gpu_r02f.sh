mkdir -p gpurun_out/r02f
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_msda_gpu.py tests/test_full_size_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r02f/pytest.log
for c in cfg2 cfg1 cfg4 cfg5 cfg3; do
  (timeout 600 python bench.py --config $c --no-cpu-baseline 2>gpurun_out/r02f/bench_$c.err) > gpurun_out/r02f/bench_$c.json
done
(timeout 900 python bench.py 2>gpurun_out/r02f/bench_default.err) > gpurun_out/r02f/bench_default.json
