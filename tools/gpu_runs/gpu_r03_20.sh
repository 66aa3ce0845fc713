# Round 3, call 20: 3x3 convolution fetch schedule: pointer loads + select (0), buffer loads (1), + two LDS stages (2)
mkdir -p gpurun_out/r03_20
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_20
for m in 0 1 2; do
  echo "## TF_CONV3_BUFLOAD=$m"
  TF_CONV3_BUFLOAD=$m timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv2|downsample|per frame"
done > $O/conv3_bufload.txt 2>&1
cat $O/conv3_bufload.txt
