mkdir -p gpurun_out/r03b
cd $GRAFT_REPO_ROOT
(timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r03b/bench.err) > gpurun_out/r03b/bench_driver_cmd.json
