mkdir -p gpurun_out/r02s
cd $GRAFT_REPO_ROOT
(timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6) > gpurun_out/r02s/pytest_gpu_all.log
(timeout 600 python bench.py 2>gpurun_out/r02s/bench_default.err) > gpurun_out/r02s/bench_default.json
for c in cfg1 cfg3 cfg4 cfg5; do
  (timeout 600 python bench.py --config $c 2>/dev/null) > gpurun_out/r02s/bench_$c.json
done
