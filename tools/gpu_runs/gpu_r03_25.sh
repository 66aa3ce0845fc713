# Round 3, call 25: 3x3 convolutions with loads issued one (1) or two (2) K-slices ahead, two split-K policies
mkdir -p gpurun_out/r03_25
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_25
for m in 1 2; do
  for pol in 768,300,8,32 384,160,8,64; do
    echo "## TF_CONV3_BUFLOAD=$m TF_CONV_KSPLIT_POLICY=$pol"
    TF_CONV3_BUFLOAD=$m TF_CONV_KSPLIT_POLICY=$pol timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv2|downsample|per frame"
  done
done > $O/conv3_ahead2.txt 2>&1
cat $O/conv3_ahead2.txt
