# Round 6, call 18: independent launches of a frame as parallel branches (runtime.Fork: bottleneck shortcut, input projections, value projection):
# parity (full-size goldens, pipelined 64 frames), then the frame with and without
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_18
mkdir -p $O
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "cfg2_full or pipelined_tracker_64 or well_conditioned_64 or track_ids_bit_exact" > $O/pytest_full.txt 2>&1; tail -3 $O/pytest_full.txt
for v in branches single branches2 single2; do
  case $v in single*) export TF_BRANCHES=0;; *) unset TF_BRANCHES;; esac
  timeout 600 python bench.py --no-cpu-baseline --no-fp32-exact --no-split3 --no-roofline --sequences 1 > $O/bench_$v.json 2> $O/bench_$v.err
  python3 -c "
import json
d=json.load(open('$O/bench_$v.json'))
print('$v', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'host', d.get('host_frames_fps'), 'parity', d['parity']['max_abs_boxes'], d['parity']['ids_equal'], d['parity']['pipelined']['frames_prepared'])"
done
