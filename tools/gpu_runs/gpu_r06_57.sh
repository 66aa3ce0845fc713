#!/bin/bash
# Round 6, call 57: the side streams of three interleaved lanes under the NARROW schedule (cfg 4, then cfg 5 on the best ones).
OUT=gpurun_out/r06_57; mkdir -p $OUT
python - <<'PY' > $OUT/cands.txt
import random
r = random.Random(5)
c = [(1,14,6), (0,1,2), (5,6,8), (8,9,10), (12,13,14), (0,5,2), (2,2,2), (9,9,9), (0,0,0), (5,5,5)]
for _ in range(16):
    c.append(tuple(r.sample([i for i in range(16) if i not in (4,3,7)], 3)))
for s in c:
    print(",".join(map(str, s)))
PY
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3 --no-single-sequence"
while read sides; do
  TF_LANE_SIDES_NARROW=$sides python bench.py --config cfg4 $FAST > $OUT/run.json 2> $OUT/run.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('cfg4 sides $sides', d['value'])
except Exception as e: print('$sides FAILED', e)" | tee -a $OUT/summary_cfg4.txt
done < $OUT/cands.txt
echo BEST; sort -k4 -n -r $OUT/summary_cfg4.txt | head -6 | tee $OUT/best_cfg4.txt
for sides in $(awk '{print $3}' $OUT/best_cfg4.txt) 1,14,6 0,1,2; do
  TF_LANE_SIDES_NARROW=$sides python bench.py --config cfg5 $FAST > $OUT/run.json 2> $OUT/run.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('cfg5 sides $sides', d['value'])
except Exception as e: print('$sides FAILED', e)" | tee -a $OUT/summary_cfg5.txt
done
