# Round 4, call 10: the other BASELINE configurations with the fp16 split product (the default), and the sequences-per-GPU sweep.
mkdir -p gpurun_out/r04_10
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_10
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
for c in cfg4 cfg5 cfg1; do
  timeout 400 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python - $c <<'PY'
import json, sys
d = json.load(open('gpurun_out/r04_10/bench_%s.json' % sys.argv[1]))
print(sys.argv[1], {k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps', 'split6_fps', 'split3_fps')}, (d.get('roofline') or {}).get('frac'))
PY
done
timeout 900 python bench.py --config cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_10/bench_cfg3.json'))
print('cfg3', {k: d.get(k) for k in ('value', 'ms_per_step')}, {k: (d.get('roofline') or {}).get(k) for k in ('kernel', 'avg_launch_us', 'frac')}, (d.get('cpu_baseline') or {}).get('value'))
PY
for s in 2 3 4 5 6; do
  echo "## --sequences $s"
  timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-parity --no-fp32-exact --no-split3 --no-single-sequence --sequences $s 2> /dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['config']['sequences_per_gpu'])"
done > $O/sequences_sweep.txt 2>&1
cat $O/sequences_sweep.txt
