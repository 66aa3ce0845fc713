# timing ablations of stream_gemm_kernel (convolution form, 128 -> 128 3 x 3): without the fp32 -> fp16-pieces split, without the MFMAs, without both
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r30
out=gpurun_out/r30/stream_gemm_ablations.txt
echo "# tools/experiments/conv3_tiles.py with variant libraries (tools/build_variant.py --source linear_stream.hip -DTF_STREAM_ABLATE=...): 1 = no activation split, 2 = no MFMAs" > $out
for v in default st_a1 st_a2 st_a3; do
  echo "== $v" >> $out
  if [ $v = default ]; then lib=trackformer_amd/lib/libtf_msda.so; else lib=tools/bin/ablate/libtf_msda_$v.so; fi
  TF_MSDA_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python tools/experiments/conv3_tiles.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
