# Round 6, call 37: the tracker's mask post-processing in one launch (tf_mask_label_map_f32): kernel + tracker tests, bench cfg5
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_37
mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_models_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "label_map or mask or cfg5 or segm" > $O/pytest_mask.txt 2>&1; tail -4 $O/pytest_mask.txt
for v in 1 0; do
TF_POSTPROCESS_FUSED=$v timeout 900 python bench.py --config cfg5 --no-cpu-baseline --no-fp32-exact --no-split3 --no-roofline > $O/bench_cfg5_$v.json 2> $O/bench_cfg5.err
python3 -c "
import json
d=json.load(open('$O/bench_cfg5_$v.json'))
print('cfg5 fused post-processing=$v', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'plain', d.get('plain_step_fps') and d['plain_step_fps']['deferred_association'], 'multi', d.get('multi_sequence_fps'))"
done
