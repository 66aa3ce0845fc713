# Round 4, call 1 (prepared at the end of round 3, when the GPU minutes were gone): time what was written after the last one.
#   (a) split_conv3_kernel with the operands read one k-step ahead (TF_CONV3_BUFLOAD=2): per-layer table
#       against the default; promote (and carry the schedule over to split_gemm_body) or delete.
#   (b) the default bench line: it includes the third host pass of the tracker (positions / scores as references) and the
#       positional-add GEMM fix, neither of which has a frames/s number yet.
mkdir -p gpurun_out/r04_01
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04_01
for m in 1 2; do
  echo "## TF_CONV3_BUFLOAD=$m"
  TF_CONV3_BUFLOAD=$m timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv2|downsample|conv1 |per frame"
done > $O/conv3_operands_ahead.txt 2>&1
cat $O/conv3_operands_ahead.txt
timeout 120 python -m pytest tests/test_full_size_gpu.py tests/test_linear_split_gpu.py -m gpu -q -x -k "conv or add or split" 2>&1 | tail -3 | tee $O/pytest_conv_add.txt
timeout 100 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_01/bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps')}, d['parity'])
PY
