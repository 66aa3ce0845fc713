# Round 3, call 19: 3x3 convolution output tile: 64x128 vs 64x64 when the wide tile leaves few workgroups
mkdir -p gpurun_out/r03_19
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_19
for nb in 0 300 600 1200; do
  echo "## TF_CONV3_NARROW_BELOW=$nb"
  TF_CONV3_NARROW_BELOW=$nb timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv2|per frame"
done > $O/conv3_tile.txt 2>&1
cat $O/conv3_tile.txt
