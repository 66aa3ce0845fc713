#!/bin/bash
# Round 6, call 52: the streams of three interleaved lanes as fixed members of the pool: which layout (TF_LANE_MAINS / TF_LANE_SIDES).
OUT=gpurun_out/r06_52; mkdir -p $OUT
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3 --no-single-sequence"
run() {
  tag=$1; shift
  env "$@" python bench.py $FAST > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json "$tag $*" <<'PY' | tee -a gpurun_out/r06_52/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value(3 lanes)", d["value"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run L1 TF_LANE_MAINS=0,1,2 TF_LANE_SIDES=4,5,6
run L2 TF_LANE_MAINS=0,1,2 TF_LANE_SIDES=5,6,7
run L3 TF_LANE_MAINS=0,1,2 TF_LANE_SIDES=3,7,11
run L4 TF_LANE_MAINS=0,4,8 TF_LANE_SIDES=1,2,3
run L5 TF_LANE_MAINS=0,1,2 TF_LANE_SIDES=6,7,4
run L6 TF_LANE_MAINS=0,1,2 TF_LANE_SIDES=1,2,3
run L7 TF_LANE_MAINS=0,2,4 TF_LANE_SIDES=1,3,5
run L8 TF_LANE_MAINS=0,4,8 TF_LANE_SIDES=12,16,20
run L9 TF_LANE_MAINS=0,4,8 TF_LANE_SIDES=1,5,9
for c in cfg5 cfg4; do
  python bench.py --config $c $FAST > $OUT/$c.json 2> $OUT/$c.err
  python -c "
import json; d=json.loads(open('$OUT/$c.json').read().strip().splitlines()[-1]); print('$c 3 lanes', d['value'])" | tee -a $OUT/summary.txt
done
