# Round 4, call 2: the six-term split product (fp32-accurate, the new default) on the hardware.
#   (a) parity: GPU tests of the GEMM family for both term counts, the full-size model / tracker goldens
#   (b) 64-frame id parity: fp32 library vs six terms vs three terms
#   (c) time: every dense kernel with three and six terms (harnesses + per-layer convolution table), the bench line with all legs
mkdir -p gpurun_out/r04_02
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04_02
timeout 600 python -m pytest tests/test_linear_split_gpu.py tests/test_fused_gpu.py tests/test_input_proj_fused.py tests/test_backbone_conv1x1.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_gemm_family.txt
timeout 600 python -m pytest tests/test_full_size_gpu.py -m gpu -q -x -k "not 64_frame" 2>&1 | tail -5 | tee $O/pytest_full_size.txt
timeout 300 python tools/id_parity_64.py --setups fp32_library,split6,split3 > $O/id_parity_64.txt 2> $O/id_parity_64.err
cat $O/id_parity_64.txt; tail -3 $O/id_parity_64.err
for t in 3 6; do
  echo "## TF_SPLIT_TERMS=$t"
  for shape in "22223 256 256" "22223 256 384" "22223 256 1024 packed" "22223 1024 256 packed" "400 256 256" "66800 64 256" "16700 512 128"; do
    TF_SPLIT_TERMS=$t timeout 60 tools/bin/linear_bench $shape 2>&1 | grep -E "max .err|us per launch"
  done
  TF_SPLIT_TERMS=$t timeout 120 tools/bin/ffn_bench 22223 1024 2>&1 | tail -12
  for ti in 1 2; do echo "# linln_ti=$ti"; TF_LINLN_TI=$ti TF_SPLIT_TERMS=$t timeout 120 tools/bin/ffn_bench 22223 1024 2>&1 | grep -i "res_ln\|projection" | tail -3; done
done > $O/harness_terms.txt 2>&1
tail -40 $O/harness_terms.txt
for t in 3 6; do
  echo "## TF_SPLIT_TERMS=$t"
  TF_SPLIT_TERMS=$t timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv|downsample|per frame|stem"
done > $O/conv_per_layer_terms.txt 2>&1
grep "per frame\|##" $O/conv_per_layer_terms.txt
timeout 280 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_02/bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps', 'single_sequence_fp32_exact_fps', 'split3_fps')}, d['parity'])
print(json.dumps(d['mfma_utilisation']['live'], indent=1))
PY
