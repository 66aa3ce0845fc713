# Round 4, call 6: the other BASELINE configurations (one JSON line each, kept under profiles/), the sequences-per-GPU sweep at six
# terms, and the pieces of the GPU suite that changed after call 5 (the stream form of the convolutions forced for every shape).
mkdir -p gpurun_out/r04_06
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_06
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_linear_split_gpu.py tests/test_bench_ranks_gpu.py tests/test_models_gpu.py -m gpu -q -x -k "conv or dispatch or 64_frame or cfg1 or cfg3 or cfg5 or ranks or training or model_matches" --durations=25 2>&1 | tail -40 | tee $O/pytest_changed.txt
for c in cfg5 cfg4 cfg1; do
  timeout 400 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python - $c <<'PY'
import json, sys
d = json.load(open('gpurun_out/r04_06/bench_%s.json' % sys.argv[1]))
print(sys.argv[1], {k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps', 'split3_fps')}, (d.get('roofline') or {}).get('frac'))
PY
done
timeout 900 python bench.py --config cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_06/bench_cfg3.json'))
print('cfg3', {k: d.get(k) for k in ('value', 'ms_per_step')}, d.get('roofline'), d.get('cpu_baseline'))
PY
for s in 1 2 4 5; do
  echo "## --sequences $s"
  timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-parity --no-fp32-exact --no-split3 --no-single-sequence --sequences $s 2> /dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['config']['sequences_per_gpu'])"
done > $O/sequences_sweep.txt 2>&1
cat $O/sequences_sweep.txt
