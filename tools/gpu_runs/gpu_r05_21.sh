cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r21
out=gpurun_out/r21/bwd_mm.txt
echo "# msda_bwd_f32_mm on MI355X: correctness (tests/test_msda_gpu.py -k 'backward or grad') and tools/bench_msda.py --no-forward" > $out
for v in 32 64; do
  echo "== correctness TF_MSDA_BWD_MM=$v" >> $out
  TF_MSDA_BWD_MM=$v timeout 600 python -m pytest tests/test_msda_gpu.py -q -k "backward or grad" 2>&1 | tail -4 >> $out
done
for spec in "32 0" "32 192" "32 128" "64 0" "64 384" "64 320" "0 0"; do
  set -- $spec
  echo "== TF_MSDA_BWD_MM=$1 TF_MSDA_BWD_MM_ROWS=$2" >> $out
  TF_MSDA_BWD_MM=$1 TF_MSDA_BWD_MM_ROWS=$2 timeout 300 python tools/bench_msda.py --shapes cfg2_encoder,cfg3_encoder_n2 --no-forward --modes local,init,uniform --iters 20 2>&1 | grep -v amdgpu.ids | tail -6 >> $out
done
cat $out
