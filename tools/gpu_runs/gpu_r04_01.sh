# Round 4, call 1: (a) VERDICT r03 item 1 -- the 64-frame reference-Tracker fixture to the end under fp32 library / bf16x3 /
# bf16x3 + fp32 heads; (b) what round 3 left untimed: split_conv3_kernel with operands read one k-step ahead
# (TF_CONV3_BUFLOAD=2) against the default; (c) the default bench line on HEAD.
mkdir -p gpurun_out/r04_01
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04_01
timeout 420 python tools/id_parity_64.py --setups fp32_library,split3,split3_heads_fp32 > $O/id_parity_64.txt 2> $O/id_parity_64.err
cat $O/id_parity_64.txt; tail -3 $O/id_parity_64.err
for m in 1 2; do
  echo "## TF_CONV3_BUFLOAD=$m"
  TF_CONV3_BUFLOAD=$m timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv2|downsample|conv1 |per frame"
done > $O/conv3_operands_ahead.txt 2>&1
tail -4 $O/conv3_operands_ahead.txt
timeout 150 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_01/bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps')}, d['parity'])
PY
