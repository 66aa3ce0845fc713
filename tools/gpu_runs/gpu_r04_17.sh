# Round 4, call 17: the closing tree once more -- GPU suite, smoke(), the default bench line.
mkdir -p gpurun_out/r04_17
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_17
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_17/bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps', 'split6_fps', 'split3_fps')})
print(d['parity']); print(d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('traffic')); print((d['cpu_baseline'] or {}).get('value'))
PY
