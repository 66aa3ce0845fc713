cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r19
timeout 900 python -m pytest tests/test_models_gpu.py -q -k "pipelined or well_conditioned" 2>&1 | tail -8 > gpurun_out/r19/tests.txt
timeout 600 python bench.py > gpurun_out/r19/bench.json 2> gpurun_out/r19/bench.err
timeout 600 python bench.py --no-prepare > gpurun_out/r19/bench_no_prepare.json 2> gpurun_out/r19/bench_np.err
cat gpurun_out/r19/tests.txt; tail -3 gpurun_out/r19/bench.err; python tools/summarize_bench.py gpurun_out/r19/bench.json 2>/dev/null | head -20 || head -c 1500 gpurun_out/r19/bench.json
