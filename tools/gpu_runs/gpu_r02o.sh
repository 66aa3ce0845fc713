mkdir -p gpurun_out/r02o
cd $GRAFT_REPO_ROOT
(for shape in "22223 256 256" "22223 256 384" "22223 256 1024"; do
  timeout 60 tools/bin/linear_bench $shape 2 2>&1
  TF_LINEAR_WS_SPLIT=1 timeout 60 tools/bin/linear_bench $shape 6 2>&1
  TF_LINEAR_WS_SPLIT=2 timeout 60 tools/bin/linear_bench $shape 6 2>&1
done) > gpurun_out/r02o/linear_ws.log
(timeout 120 python tools/bench_mha.py 2>&1 | tail -2) > gpurun_out/r02o/mha.log
