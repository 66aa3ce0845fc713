# Round 3, call 26: the few-pixel 1x1 convolutions (layer3 / layer4 conv1, strided projections) with their K loop split
mkdir -p gpurun_out/r03_26
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_26
for m in 0 1; do
  echo "## TF_CONV1X1_SPLITK=$m"
  TF_CONV1X1_SPLITK=$m timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv1|downsample|per frame"
done > $O/conv1x1_splitk.txt 2>&1
cat $O/conv1x1_splitk.txt
