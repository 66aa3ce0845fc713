# Round 4, call 8: the fp16 split product with the third weight operand derived in registers (two stored pieces): harnesses,
# row-tile sweeps, GPU tests, id parity, convolution table, bench.
mkdir -p gpurun_out/r04_08
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_08
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
{
for T in 16 3; do
  echo "## TF_SPLIT_TERMS=$T"
  for s in "22223 256 256" "22223 256 384" "22223 256 256 packed" "22223 256 1024 packed" "22223 1024 256 packed" "400 256 256" "66800 64 256" "66800 64 256 packed" "66800 64 64" "66800 64 64 packed" "16700 512 128 packed" "16700 512 128" "16700 128 512" "16700 128 512 packed"; do
    echo "# $s"; TF_SPLIT_TERMS=$T timeout 60 tools/bin/linear_bench $s 2>&1 | grep -E "us per launch|checked|differ"
  done
  TF_SPLIT_TERMS=$T timeout 90 tools/bin/ffn_bench 22223 1024 2>&1 | grep -E "fused|differ|max"
done
for ti in 1 2 3; do echo "## fp16 pieces, TF_LINLN_TI=$ti"; TF_SPLIT_TERMS=16 TF_LINLN_TI=$ti timeout 90 tools/bin/ffn_bench 22223 1024 2>&1 | grep -E "separate \(packed linear"; done
for ti in 2 3; do echo "## fp16 pieces, TF_FFN_TI=$ti"; TF_SPLIT_TERMS=16 TF_FFN_TI=$ti timeout 90 tools/bin/ffn_bench 22223 1024 2>&1 | grep -E "separate \(linear1"; done
for ti in 2 3 4; do echo "## fp16 pieces, stream TI=$ti"; for s in "22223 256 1024" "22223 1024 256" "22223 256 256"; do TF_SPLIT_TERMS=16 timeout 60 tools/bin/linear_bench $s packed$ti 2>&1 | grep -E "us per launch"; done; done
echo "## hidden 288 (cfg 4), fp16 pieces"; TF_SPLIT_TERMS=16 timeout 90 tools/bin/ffn_bench 30000 1024 288 2>&1 | grep -E "fused|differ|max"
} > $O/harness_terms.txt 2>&1
grep -E "##|# |us per launch|fused:" $O/harness_terms.txt | cut -c1-110
timeout 600 python -m pytest tests/test_linear_split_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -k "fp16 or f16 or 64_frame" -s 2>&1 | grep -E "passed|failed|Error|64-frame fixture|max \|d|conv \(" | tail -40 | tee $O/pytest_fp16.txt
timeout 400 python tools/id_parity_64.py --frames 64 --setups fp32_library,split16 > $O/id_parity_64.txt 2> $O/id_parity_64.err
tail -4 $O/id_parity_64.txt
{
for T in 16; do for S in 0 all; do for P in 0 1; do
  echo "## TF_SPLIT_TERMS=$T TF_CONV_STREAM=$S TF_LINEAR_PACKED=$P"
  TF_SPLIT_TERMS=$T TF_CONV_STREAM=$S TF_LINEAR_PACKED=$P timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv|downsample|per frame"
done; done; done
} > $O/conv_per_layer.txt 2>&1
grep -E "##|per frame" $O/conv_per_layer.txt
timeout 400 python bench.py --no-cpu-baseline --split-terms 16 --no-fp32-exact > $O/bench_f16.json 2> $O/bench_f16.err
python - <<'PY'
import json
for n in ('bench_f16',):
    try:
        d = json.load(open('gpurun_out/r04_08/%s.json' % n))
    except Exception as e:
        print(n, 'unreadable', e); continue
    print(n, {k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'split6_fps', 'split_f16_fps', 'split3_fps')}, d.get('parity'))
    print(json.dumps((d.get('mfma_utilisation') or {}).get('live'))[:1200])
PY
