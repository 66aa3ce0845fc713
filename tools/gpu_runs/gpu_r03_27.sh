# Round 3, call 27 (closing validation): bench.py as the driver runs it, the GPU suite (one of the two training-step tests
# left out for the GPU-minute budget: the mask variant covers the same path plus the mask losses), smoke()
mkdir -p gpurun_out/r03_27
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03_27
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r03_27/bench_default.json'))
    for k in ('value','ms_per_step','single_sequence_fps','multi_sequence_fps','fp32_exact_fps','association','cpu_baseline'):
        print(k, d.get(k))
    print(d['parity'], d['roofline'])
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -3 $O/bench_default.err
timeout 420 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider \
  --deselect tests/test_models_gpu.py::test_training_step_matches_reference_cpu_path 2>&1 | tail -25 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
