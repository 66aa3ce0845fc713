# Round 6, call 25: (a) cfg 3 with the training fold and no publish barriers inside a step: default / library convolutions for the trainable
# layers of the no-grad pass / off; (b) the multi-frame model (cfg 4) with the image-only half prepared ahead: parity tests, bench with and
# without; (c) the mask tracker test
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_25
mkdir -p $O
for v in 1 lib 0; do
  TF_TRAIN_FOLD=$v timeout 600 python bench.py --config cfg3 --no-cpu-baseline --no-roofline > $O/bench_cfg3_fold_$v.json 2> $O/bench_cfg3_fold_$v.err
  python3 -c "
import json
d=json.load(open('$O/bench_cfg3_fold_$v.json')); print('cfg3 fold=$v', d['value'], d['ms_per_step'])"
done
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "multi_frame or cfg4 or mask or training_fold" > $O/pytest_multi.txt 2>&1; tail -5 $O/pytest_multi.txt
for v in prepare noprepare; do
  extra=""; [ $v = noprepare ] && extra="--no-prepare"
  timeout 900 python bench.py --config cfg4 --no-cpu-baseline --no-fp32-exact --no-split3 $extra > $O/bench_cfg4_$v.json 2> $O/bench_cfg4_$v.err
  python3 -c "
import json
d=json.load(open('$O/bench_cfg4_$v.json'))
print('cfg4 $v', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'multi', d.get('multi_sequence_fps'), 'parity', d.get('parity') and {k: d['parity'].get(k) for k in ('ids_equal','max_abs_boxes','path')})"
done
