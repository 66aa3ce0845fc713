# four register stages of activations ahead in the convolution form of the stream GEMM (the default now) against two (variant library)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r33
out=gpurun_out/r33/stream_xdepth.txt
: > $out
for v in default st_x2; do
  echo "== $v (TF_STREAM_XDEPTH = $([ $v = default ] && echo 4 || echo 2))" >> $out
  if [ $v = default ]; then lib=trackformer_amd/lib/libtf_msda.so; else lib=tools/bin/ablate/libtf_msda_$v.so; fi
  TF_MSDA_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python tools/experiments/conv3_tiles.py 2>&1 | grep -v amdgpu.ids | grep -E "rows|16384  2|16896  2|131072  1|32768  1" >> $out
  TF_MSDA_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/bench_conv.py --iters 20 2>&1 | grep -v amdgpu.ids | tail -32 >> $out
done
cat $out
