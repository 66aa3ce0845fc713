#!/bin/bash
# Round 6, call 58: the sequence streams of three interleaved lanes under the NARROW schedule (cfg 4: no look-ahead with several lanes,
# only these streams matter), the best ones on cfg 5.
OUT=gpurun_out/r06_58; mkdir -p $OUT
python - <<'PY' > $OUT/cands.txt
import random
r = random.Random(3)
c = [(4,3,7), (0,1,2), (0,4,8), (1,2,3), (0,2,5), (5,6,7), (8,9,10), (0,1,3), (2,3,4)]
for _ in range(13):
    c.append(tuple(r.sample(range(16), 3)))
for s in c:
    print(",".join(map(str, s)))
PY
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3 --no-single-sequence"
while read mains; do
  TF_LANE_MAINS_NARROW=$mains python bench.py --config cfg4 $FAST > $OUT/run.json 2> $OUT/run.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('cfg4 mains $mains', d['value'])
except Exception as e: print('$mains FAILED', e)" | tee -a $OUT/summary_cfg4.txt
done < $OUT/cands.txt
echo BEST; sort -k4 -n -r $OUT/summary_cfg4.txt | head -5 | tee $OUT/best_cfg4.txt
for mains in $(awk '{print $3}' $OUT/best_cfg4.txt) 4,3,7; do
  for sides in 1,14,6 12,8,14; do
  TF_LANE_MAINS_NARROW=$mains TF_LANE_SIDES_NARROW=$sides python bench.py --config cfg5 $FAST > $OUT/run.json 2> $OUT/run.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('cfg5 mains $mains sides $sides', d['value'])
except Exception as e: print('$mains FAILED', e)" | tee -a $OUT/summary_cfg5.txt
  done
done
