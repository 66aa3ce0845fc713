# Round 6, call 28: the tracker's post-processing in one launch (tf_postprocess_pack_f32), the multi-frame prepare edge cases: the whole
# -m gpu suite, then the default bench with the fused post-processing on and off
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_28
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu_all.txt 2>&1; tail -6 $O/pytest_gpu_all.txt
for v in 1 0; do
  TF_POSTPROCESS_FUSED=$v timeout 900 python bench.py --no-fp32-exact --no-split3 --no-cpu-baseline --no-roofline > $O/bench_post_$v.json 2> $O/bench_post_$v.err
  python3 - <<PY
import json
d=json.load(open('$O/bench_post_$v.json'))
print('postprocess fused=$v: value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'host', d.get('host_frames_fps'), 'plain', d.get('plain_step_fps') and (d['plain_step_fps']['deferred_association'], d['plain_step_fps']['association_before_return']), 'multi', d.get('multi_sequence_fps'), 'parity ids', d['parity']['ids_equal'], d['parity']['pipelined']['ids_equal'])
PY
done
