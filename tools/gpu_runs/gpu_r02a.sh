set -x
mkdir -p gpurun_out/r02a
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q -s 2>&1 | tail -40) > gpurun_out/r02a/pytest_full.log
(timeout 300 tools/bin/msda_bench --iters 20 --sets 4 --patterns init,pert,local direct quad quad:npass=1 quad:npass=2 quad:ta=12 2>&1) > gpurun_out/r02a/quad_sweep_cold.log
(timeout 200 tools/bin/msda_bench --iters 20 --sets 1 --patterns init,pert direct quad 2>&1) > gpurun_out/r02a/quad_sweep_warm.log
(timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r02a/bench_default.err) > gpurun_out/r02a/bench_default.json
(timeout 300 python bench.py --no-cpu-baseline --no-roofline --split-linear 2>gpurun_out/r02a/bench_split.err) > gpurun_out/r02a/bench_split.json
(timeout 300 python bench.py --no-cpu-baseline --no-roofline --sequences 1 2>/dev/null) > gpurun_out/r02a/bench_seq1.json
(timeout 300 python bench.py --no-cpu-baseline --no-roofline --sequences 1 --split-linear 2>/dev/null) > gpurun_out/r02a/bench_seq1_split.json
