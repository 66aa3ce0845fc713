# Round 5, call 12: the pipelined tracker (step_prepare) -- GPU tests of the model / tracker files, then the bench line
mkdir -p gpurun_out/r05_12
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_12
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_full_size_gpu.py tests/test_msda_gpu.py -m gpu -q 2>&1 | tail -8 | tee $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_12/bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'multi_sequence_fps', 'fp32_exact_fps', 'single_sequence_fp32_exact_fps')})
print(d['parity']); print(d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline']['kernel'][:40])
PY
timeout 300 python bench.py --no-cpu-baseline --no-prepare --no-roofline --no-parity --no-fp32-exact --no-split3 > $O/bench_noprepare.json 2> $O/bench_noprepare.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_12/bench_noprepare.json'))
print('without step_prepare', {k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'multi_sequence_fps')})
PY
