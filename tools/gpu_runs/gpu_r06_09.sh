# Round 6, call 9: timing ablations of the halo convolution kernel: 1 = weights from L1 (one 4 KB fragment set re-read), 2 = halo staged once,
# 4 = no MFMAs, 5 = 1 + 4
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_09
mkdir -p $O
for v in default halo_a1 halo_a2 halo_a4 halo_a5; do
  if [ $v = default ]; then unset TF_MSDA_LIB; else export TF_MSDA_LIB=$GRAFT_REPO_ROOT/tools/bin/ablate/libtf_msda_$v.so; fi
  echo "== $v"
  timeout 400 python tools/bench_conv.py --iters 20 2>&1 | grep -v amdgpu.ids | grep -E "conv2 " | grep " 3 1 " | cut -c1-100
done
