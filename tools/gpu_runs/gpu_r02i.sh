mkdir -p gpurun_out/r02i
cd $GRAFT_REPO_ROOT
(for shape in "22223 256 256" "22223 256 384" "22223 256 1024" "22223 1024 256" "400 256 256"; do
  for v in 0 1 2 3 4 5; do timeout 60 tools/bin/linear_bench $shape $v 2>&1; done
done) > gpurun_out/r02i/linear_variants.log
(timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_linear_split_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r02i/pytest_fused.log
