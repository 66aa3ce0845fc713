mkdir -p gpurun_out/r02n
cd $GRAFT_REPO_ROOT
(for shape in "22223 256 256" "22223 256 384" "22223 256 1024" "21760 256 256" "5000 256 128"; do
  for v in 2 6; do timeout 60 tools/bin/linear_bench $shape $v 2>&1; done
done) > gpurun_out/r02n/linear_ws.log
(timeout 120 python tools/bench_mha.py 2>&1 | tail -2) > gpurun_out/r02n/mha.log
