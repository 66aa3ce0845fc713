# Round 6, call 39: the mask head's levels with the FPN merge and the previous GroupNorm + ReLU in the convolution's fetch
# (tf_conv3x3_merge_packed_f32): kernel + mask tests, bench cfg5
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_39
mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_models_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "merged or label_map or upsample_add or one_channel or mask or cfg5 or segm" > $O/pytest_mask.txt 2>&1; tail -4 $O/pytest_mask.txt
timeout 900 python bench.py --config cfg5 --no-cpu-baseline --no-fp32-exact --no-split3 --no-roofline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python3 -c "
import json
d=json.load(open('$O/bench_cfg5.json'))
print('cfg5', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'plain', d.get('plain_step_fps') and d['plain_step_fps']['deferred_association'], 'multi', d.get('multi_sequence_fps'))"
