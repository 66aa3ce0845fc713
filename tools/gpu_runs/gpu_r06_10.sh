# Round 6, call 10: the split-K policy of the 3 x 3 layers under the halo form (target blocks, leave-alone blocks, min slices per piece, min slices)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_10
mkdir -p $O
for pol in 768,300,8,32 768,200,8,32 768,100,8,32 768,50,8,32 1536,300,8,32 1536,300,18,32 1024,300,9,32 512,300,9,32; do
  echo "== TF_CONV_KSPLIT_POLICY=$pol"
  TF_CONV_KSPLIT_POLICY=$pol timeout 400 python tools/bench_conv.py --iters 20 2>&1 | grep -v amdgpu.ids | grep -E "conv2 |per frame" | grep -E " 3 1 |per frame" | cut -c1-100
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/bench_conv.py --iters 5 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/conv_kernel_stats.csv; rm -rf $O/prof
head -30 $O/conv_kernel_stats.csv | cut -c1-60,200-330
