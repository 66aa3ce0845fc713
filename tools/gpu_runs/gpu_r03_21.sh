# Round 3, call 21: split-K policy of the 3x3 convolutions (target workgroups, leave-alone blocks, slices per piece, min slices)
mkdir -p gpurun_out/r03_21
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_21
for pol in 384,160,8,64 768,300,8,32 1024,300,6,32 1024,600,4,16 2048,600,4,16; do
  for m in 1 2; do
    echo "## TF_CONV_KSPLIT_POLICY=$pol TF_CONV3_BUFLOAD=$m"
    TF_CONV_KSPLIT_POLICY=$pol TF_CONV3_BUFLOAD=$m timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv2|per frame"
  done
done > $O/conv3_ksplit.txt 2>&1
cat $O/conv3_ksplit.txt
