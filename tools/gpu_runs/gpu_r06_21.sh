# Round 6, call 21: the mask-head model (cfg 5) with the image-only half prepared ahead (two graphs): parity of the mask tracker, bench cfg5
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_21
mkdir -p $O
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "mask or segm or cfg5" > $O/pytest_mask.txt 2>&1; tail -3 $O/pytest_mask.txt
for v in prepare noprepare; do
  extra=""; [ $v = noprepare ] && extra="--no-prepare"
  timeout 900 python bench.py --config cfg5 --no-cpu-baseline --no-fp32-exact --no-split3 --no-roofline --sequences 1 $extra > $O/bench_cfg5_$v.json 2> $O/bench_cfg5_$v.err
  python3 -c "
import json
d=json.load(open('$O/bench_cfg5_$v.json'))
print('$v', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'))"
done
