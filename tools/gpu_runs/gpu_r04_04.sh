# Round 4, call 4: the stream GEMM's new forms (residual epilogue, narrow blocks, convolutions), the tail split of the one-launch
# blocks: parity on the hardware, then A/B timing against the block kernels, then the bench line.
mkdir -p gpurun_out/r04_04
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04_04
timeout 900 python -m pytest tests/test_linear_split_gpu.py tests/test_fused_gpu.py tests/test_input_proj_fused.py tests/test_backbone_conv1x1.py tests/test_bench_ranks_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_gemm_family.txt
timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -q -x 2>&1 | tail -12 | tee $O/pytest_full_size.txt
timeout 300 python -m pytest tests/test_models_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest_models.txt
{
for split in 0 1; do
  echo "## six terms, TF_FFN_TAIL_SPLIT=$split"
  TF_FFN_TAIL_SPLIT=$split TF_SPLIT_TERMS=6 timeout 120 tools/bin/ffn_bench 22223 1024 2>&1 | grep -E "separate|fused:"
done
for ti in 1 2; do echo "## six terms, TF_LINLN_TI=$ti"; TF_LINLN_TI=$ti TF_SPLIT_TERMS=6 timeout 120 tools/bin/ffn_bench 22223 1024 2>&1 | grep -E "packed linear, residual"; done
echo "## hidden 288 (cfg 4), six terms, tail split 0 / 1"
for split in 0 1; do TF_FFN_TAIL_SPLIT=$split TF_SPLIT_TERMS=6 timeout 120 tools/bin/ffn_bench 44446 1024 0 288 2>&1 | grep -E "separate|fused:"; done
} > $O/one_launch_blocks_tail_split.txt 2>&1
cat $O/one_launch_blocks_tail_split.txt
for t in 6 3; do
  for cs in 0 1; do
    echo "## TF_SPLIT_TERMS=$t TF_CONV_STREAM=$cs TF_LINEAR_PACKED=$cs"
    TF_SPLIT_TERMS=$t TF_CONV_STREAM=$cs TF_LINEAR_PACKED=$cs timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv|downsample|per frame"
  done
done > $O/conv_per_layer_stream_vs_block.txt 2>&1
grep "per frame\|##" $O/conv_per_layer_stream_vs_block.txt
timeout 280 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_04/bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps', 'single_sequence_fp32_exact_fps', 'split3_fps')}, d['parity'])
print(json.dumps(d['mfma_utilisation']['live'], indent=1))
PY
