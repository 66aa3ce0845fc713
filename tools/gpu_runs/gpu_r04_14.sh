R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r04_14
for A in 0 1; do
  echo "## TF_STREAM_ABLATE=$A"
  TF_STREAM_ABLATE=$A timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv2|per frame"
  TF_STREAM_ABLATE=$A timeout 60 tools/bin/linear_bench 22223 1024 256 packed 2>&1 | grep -E "us per launch"
  TF_STREAM_ABLATE=$A timeout 60 tools/bin/linear_bench 22223 256 1024 packed 2>&1 | grep -E "us per launch"
done > gpurun_out/r04_14/ablate_split.txt 2>&1
cut -c1-110 gpurun_out/r04_14/ablate_split.txt | sed 's/ = .*TFLOP.*//'
