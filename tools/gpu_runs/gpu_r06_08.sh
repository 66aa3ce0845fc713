# Round 6, call 8: halo form with the hand-pipelined tap loop (A fragments one step ahead, weight fragments two taps ahead) against the stream form
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_08
mkdir -p $O
for v in halo stream; do
  if [ $v = stream ]; then export TF_CONV_HALO=0; else unset TF_CONV_HALO; fi
  timeout 400 python tools/bench_conv.py --iters 20 2>&1 | grep -v amdgpu.ids > $O/conv_$v.txt
  echo "== $v"; grep -E "conv2|per frame" $O/conv_$v.txt | cut -c1-120
done
unset TF_CONV_HALO
for ti in 1 2 4; do
  echo "== halo, TF_LINEAR_STREAM_TI=$ti"
  TF_LINEAR_STREAM_TI=$ti timeout 400 python tools/bench_conv.py --iters 20 2>&1 | grep -v amdgpu.ids | grep -E "conv2 " | grep " 3 1 " | cut -c1-120
done
