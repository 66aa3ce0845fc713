# Round 3, call 29 (last GPU seconds of the round): bench after the tracker's host-side pass, then the tracker GPU tests
mkdir -p gpurun_out/r03_29
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03_29
timeout 55 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-fp32-exact > $O/bench_host_pass.json 2> $O/bench_host_pass.err
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r03_29/bench_host_pass.json'))
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'association')})
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -2 $O/bench_host_pass.err
timeout 60 python -m pytest tests/test_models_gpu.py -m gpu -q -x -p no:cacheprovider -k "tracker and not mask_head and not lazy" 2>&1 | tail -4 | tee $O/pytest_tracker.txt
