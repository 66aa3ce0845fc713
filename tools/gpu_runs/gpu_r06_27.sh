# Round 6, call 27: the deferred association of Tracker.step (the reference's own loop runs the pipelined schedule): parity at both sizes,
# the affected tests, bench default (plain_step_fps in the line) and cfg 4 / cfg 5 / cfg 3
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_27
mkdir -p $O
timeout 1500 python -m pytest tests/test_models_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "unobserved or pipelined or prepared_ahead or one_device_to_host or mask or tracker" > $O/pytest_tracker.txt 2>&1; tail -5 $O/pytest_tracker.txt
timeout 900 python bench.py --no-fp32-exact --no-split3 > $O/bench_default.json 2> $O/bench_default.err
python3 - <<PY
import json
d=json.load(open('$O/bench_default.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'host', d.get('host_frames_fps'), 'plain', d.get('plain_step_fps'), 'multi', d.get('multi_sequence_fps'))
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us']); print('parity', d.get('parity') and d['parity'].get('ids_equal'), d['parity']['pipelined']['ids_equal'], d['parity']['pipelined']['frames_prepared'])
PY
for c in cfg4 cfg5; do
  timeout 900 python bench.py --config $c --no-cpu-baseline --no-fp32-exact --no-split3 > $O/bench_$c.json 2> $O/bench_$c.err
  python3 -c "
import json
d=json.load(open('$O/bench_$c.json'))
print('$c', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'host', d.get('host_frames_fps'), 'plain', d.get('plain_step_fps'), 'multi', d.get('multi_sequence_fps'))"
done
timeout 600 python bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python3 -c "
import json
d=json.load(open('$O/bench_cfg3.json')); print('cfg3', d['value'], d['ms_per_step'], d['roofline'] and d['roofline'].get('frac'))"
