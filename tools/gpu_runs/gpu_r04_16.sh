# Round 4, call 16: more independent accumulators per wave in the convolution form (TF_LINEAR_STREAM_TI forces the row tiles).
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r04_16
for T in 0 2 4; do
  echo "## TF_LINEAR_STREAM_TI=$T"
  TF_LINEAR_STREAM_TI=$T timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv|downsample|per frame"
done > gpurun_out/r04_16/conv_ti.txt 2>&1
grep -E "##|per frame|conv2|conv1 |downsample" gpurun_out/r04_16/conv_ti.txt | cut -c1-100
