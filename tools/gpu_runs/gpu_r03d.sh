# packed FFN linears on by default: goldens at full size, the linear tests, a short bench line
mkdir -p gpurun_out/r03d
cd $GRAFT_REPO_ROOT
timeout 240 python3 -m pytest tests/test_linear_split_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -s > gpurun_out/r03d/pytest.txt 2>&1
tail -3 gpurun_out/r03d/pytest.txt
(timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r03d/bench.err) > gpurun_out/r03d/bench.json
cat gpurun_out/r03d/bench.json | cut -c1-400
