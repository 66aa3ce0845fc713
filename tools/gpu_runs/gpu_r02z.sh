mkdir -p gpurun_out/r02z
cd $GRAFT_REPO_ROOT
(timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -4) > gpurun_out/r02z/pytest_gpu_all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r02z/smoke.log
(timeout 600 python bench.py --no-cpu-baseline 2>/dev/null) > gpurun_out/r02z/bench_default_nocpu.json
(timeout 600 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null) > gpurun_out/r02z/bench_cfg4_nocpu.json
