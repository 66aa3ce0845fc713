#!/bin/bash
# Round 6, call 56: the lanes' stream layout again, now that the binding of streams to hardware queues is fixed (runtime.bind_streams):
# structured + seeded random candidates, three lanes of cfg 2.
OUT=gpurun_out/r06_56; mkdir -p $OUT
python - <<'PY' > $OUT/cands.txt
import random
r = random.Random(11)
c = [((0,1,2),(3,7,11)), ((0,1,2),(7,11,15)), ((0,4,8),(1,2,3)), ((0,4,8),(1,5,9)), ((0,1,2),(4,5,6)), ((0,1,2),(5,6,4)), ((0,1,2),(6,4,5)),
     ((0,1,2),(8,9,10)), ((0,1,2),(12,13,14)), ((1,2,3),(4,8,12)), ((1,2,3),(5,6,7)), ((0,2,1),(12,8,14)), ((4,5,6),(12,8,14)), ((0,1,2),(12,8,10))]
for _ in range(34):
    mains = tuple(r.sample(range(8), 3))
    sides = tuple(r.sample([i for i in range(16) if i not in mains], 3))
    c.append((mains, sides))
for m, s in c:
    print(",".join(map(str, m)), ",".join(map(str, s)))
PY
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3 --no-single-sequence"
while read mains sides; do
  TF_LANE_MAINS=$mains TF_LANE_SIDES=$sides python bench.py $FAST > $OUT/run.json 2> $OUT/run.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('mains $mains sides $sides', d['value'])
except Exception as e: print('$mains $sides FAILED', e)" | tee -a $OUT/summary.txt
done < $OUT/cands.txt
echo BEST; sort -k5 -n -r $OUT/summary.txt | head -8
