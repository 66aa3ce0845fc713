# Round 4, call 7: the fp16 split product (terms = 16) on the hardware -- kernel harnesses against six terms, its GPU tests, the
# 64-frame id parity of every arithmetic, the per-layer convolution table, and the bench line with all legs.
mkdir -p gpurun_out/r04_07
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_07
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
{
for T in 6 16 3; do
  echo "## TF_SPLIT_TERMS=$T"
  for s in "22223 256 256" "22223 256 384" "22223 256 256 packed" "22223 256 1024 packed" "22223 1024 256 packed" "400 256 256" "66800 64 256" "66800 64 64 packed" "16700 512 128 packed" "16700 512 128"; do
    TF_SPLIT_TERMS=$T timeout 60 tools/bin/linear_bench $s 2>&1 | grep -E "us per launch|checked|differ"
  done
  TF_SPLIT_TERMS=$T timeout 90 tools/bin/ffn_bench 22223 1024 2>&1 | grep -E "fused|differ|max"
done
for ti in 1 2; do echo "## fp16 pieces, TF_LINLN_TI=$ti"; TF_SPLIT_TERMS=16 TF_LINLN_TI=$ti timeout 90 tools/bin/ffn_bench 22223 1024 2>&1 | grep -E "separate \(packed linear"; done
for ti in 2 3 4; do echo "## fp16 pieces, stream TI=$ti"; TF_SPLIT_TERMS=16 timeout 60 tools/bin/linear_bench 22223 256 1024 packed$ti 2>&1 | grep -E "us per launch"; TF_SPLIT_TERMS=16 timeout 60 tools/bin/linear_bench 22223 1024 256 packed$ti 2>&1 | grep -E "us per launch"; done
} > $O/harness_terms.txt 2>&1
tail -60 $O/harness_terms.txt
timeout 600 python -m pytest tests/test_linear_split_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -k "fp16 or f16 or 64_frame" -s 2>&1 | grep -E "passed|failed|Error|64-frame fixture|max \|d|conv \(" | tail -40 | tee $O/pytest_fp16.txt
timeout 400 python tools/id_parity_64.py --frames 64 > $O/id_parity_64.txt 2> $O/id_parity_64.err
tail -8 $O/id_parity_64.txt
{
for T in 6 16; do for S in 0 all; do
  echo "## TF_SPLIT_TERMS=$T TF_CONV_STREAM=$S"
  TF_SPLIT_TERMS=$T TF_CONV_STREAM=$S timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv|downsample|per frame"
done; done
} > $O/conv_per_layer.txt 2>&1
grep -E "##|per frame" $O/conv_per_layer.txt
timeout 500 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --no-cpu-baseline --split-terms 16 --no-split3 --no-fp32-exact --no-roofline > $O/bench_f16.json 2> $O/bench_f16.err
python - <<'PY'
import json
for n in ('bench_default', 'bench_f16'):
    try:
        d = json.load(open('gpurun_out/r04_07/%s.json' % n))
    except Exception as e:
        print(n, 'unreadable', e); continue
    print(n, {k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps', 'single_sequence_fp32_exact_fps', 'split6_fps', 'split_f16_fps', 'split3_fps')}, d.get('parity'))
    print(json.dumps((d.get('mfma_utilisation') or {}).get('live'), indent=0)[:1500])
PY
