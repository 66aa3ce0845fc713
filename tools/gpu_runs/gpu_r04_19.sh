# Round 4, call 19: the mask head through the split-product convolution kernels (cfg 5): parity tests + the step with and without.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_19
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 400 python -m pytest tests/test_full_size_gpu.py tests/test_models_gpu.py -m gpu -q -x -k "cfg5 or mask or segm" -s 2>&1 | grep -E "passed|failed|Error|cfg5_full" | tail -8 | tee $O/pytest_cfg5.txt
for S in 1 0; do
  TF_MASK_HEAD_SPLIT=$S timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-roofline --no-parity --no-fp32-exact --no-split3 2> $O/bench_cfg5_$S.err > $O/bench_cfg5_$S.json
  python - $S <<'PY'
import json, sys
d = json.load(open('gpurun_out/r04_19/bench_cfg5_%s.json' % sys.argv[1]))
print('TF_MASK_HEAD_SPLIT=%s' % sys.argv[1], {k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps')}, d.get('association'))
PY
done
