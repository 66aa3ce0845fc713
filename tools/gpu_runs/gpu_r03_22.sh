# Round 3, call 22: 128-row output tiles of the 3x3 convolutions x split-K policy
mkdir -p gpurun_out/r03_22
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_22
for t in 0 1; do
  for pol in 768,300,8,32 1024,400,8,32 1536,600,6,16; do
    echo "## TF_CONV3_TILE=$t TF_CONV_KSPLIT_POLICY=$pol"
    TF_CONV3_TILE=$t TF_CONV_KSPLIT_POLICY=$pol timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv2|downsample|per frame"
  done
done > $O/conv3_tile128.txt 2>&1
cat $O/conv3_tile128.txt
