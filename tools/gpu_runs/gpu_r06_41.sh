# Round 6, call 41: the criterion's losses for all decoder layers at once (criterion._layers_at_once): training goldens, stage times, cfg-3 line
# with and without (two runs each, alternating: the box's first training process has been seen slower)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_41
mkdir -p $O
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "train or cfg3 or loss" > $O/pytest_train.txt 2>&1; tail -3 $O/pytest_train.txt
timeout 600 python tools/train_profile.py --steps 6 2>/dev/null > $O/train_profile.txt; head -12 $O/train_profile.txt
for v in 1 0 1 0; do
  TF_CRITERION_LAYERS_AT_ONCE=$v timeout 600 python bench.py --config cfg3 --no-cpu-baseline --no-roofline > $O/bench_cfg3_$v.json 2> $O/bench.err
  python3 -c "
import json
d=json.load(open('$O/bench_cfg3_$v.json')); print('cfg3 layers at once=$v', d['value'], d['ms_per_step'])"
done
