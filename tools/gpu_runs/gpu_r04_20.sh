# Round 4, call 20 (the last seconds of the round's GPU budget): the mask-head route with query chunks -- the full-size parity test
# and one short bench leg.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_20
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 70 python -m pytest tests/test_full_size_gpu.py -m gpu -q -x -k "test_cfg5_mask_head_800x1333" -s 2>&1 | grep -E "passed|failed|Error|cfg5_full" | tail -4 | tee $O/pytest_cfg5.txt
timeout 75 python bench.py --config cfg5 --no-cpu-baseline --no-roofline --no-parity --no-fp32-exact --no-split3 --no-single-sequence --steps 6 --warmup 3 --min-seconds 0.5 2> $O/bench_cfg5.err > $O/bench_cfg5.json
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r04_20/bench_cfg5.json'))
    print('mask head through the split kernels:', {k: d.get(k) for k in ('value', 'ms_per_step')})
except Exception as e:
    print('bench unreadable', e)
PY
tail -3 $O/bench_cfg5.err | cut -c1-200
