mkdir -p gpurun_out/r02q
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r02q/pytest_gpu_all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4) > gpurun_out/r02q/smoke.log
for c in cfg5 cfg4; do
  (timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null) > gpurun_out/r02q/bench_$c.json
done
