# Round 3, call 18: bench after lazy results / calibration on the object queries; tracker tests
mkdir -p gpurun_out/r03_18
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03_18
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_18/bench_default.json'))
for k in ('value','ms_per_step','single_sequence_fps','multi_sequence_fps','fp32_exact_fps','association'):
    print(k, d.get(k))
print(d['parity']['ids_equal'], d['roofline']['avg_launch_us'])
PY
for seq in 2 4; do timeout 200 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-fp32-exact --no-single-sequence --sequences $seq > $O/bench_seq$seq.json 2> $O/bench_seq$seq.err; done
python tools/summarize_bench.py $O
timeout 300 python -m pytest tests/test_models_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -k "tracker" 2>&1 | tail -3
