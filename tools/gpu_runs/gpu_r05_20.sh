cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r20
out=gpurun_out/r20/bwd_ablations.txt
echo "# tools/bench_msda.py --shapes cfg2_encoder,cfg3_encoder_n2 --no-forward: msda_bwd_f32_sorted2 and its timing ablations (tools/build_variant.py --source msda_hip.hip)" > $out
for v in default bwd_a1 bwd_a2 bwd_a3 bwd_a4 bwd_p3 bwd_p4; do
  echo "== $v" >> $out
  if [ $v = default ]; then lib=trackformer_amd/lib/libtf_msda.so; else lib=tools/bin/ablate/libtf_msda_$v.so; fi
  TF_MSDA_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/bench_msda.py --shapes cfg2_encoder,cfg3_encoder_n2 --no-forward --modes local,init --iters 20 2>&1 | grep -v amdgpu.ids | tail -6 >> $out
done
for v in bwd_p3 bwd_p4; do
  echo "== correctness $v" >> $out
  TF_MSDA_LIB=$GRAFT_REPO_ROOT/tools/bin/ablate/libtf_msda_$v.so timeout 600 python -m pytest tests/test_msda_gpu.py -q -k "backward or grad" 2>&1 | tail -3 >> $out
done
cat $out
