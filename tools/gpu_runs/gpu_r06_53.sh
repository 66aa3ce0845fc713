#!/bin/bash
# Round 6, call 53: a seeded random search over the pool indices of three interleaved lanes' streams (NARROW and WIDE schedules).
OUT=gpurun_out/r06_53; mkdir -p $OUT
python - <<'PY' > $OUT/cands.txt
import random
r = random.Random(7)
c = []
for mains in [(0,1,2),(0,4,8),(0,8,16),(1,2,3),(0,2,5)]:
    for _ in range(7):
        pool = [i for i in range(32) if i not in mains]
        s = r.sample(pool, 3)
        c.append(("N", mains, s))
for mains in [(0,1,2),(0,4,8)]:
    for _ in range(5):
        pool = [i for i in range(16) if i not in mains]
        s = r.sample(pool, 3)
        c.append(("W", mains, s))
for k, m, s in c:
    print(k, ",".join(map(str, m)), ",".join(map(str, s)))
PY
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3 --no-single-sequence"
while read kind mains sides; do
  if [ $kind = W ]; then EXTRA="TF_GRAPH_SLOTS=4 TF_GRAPH_SIDE_STREAMS=2 TF_GRAPH_LOOKAHEAD=2"; else EXTRA="A=1"; fi
  env $EXTRA TF_LANE_MAINS=$mains TF_LANE_SIDES=$sides python bench.py $FAST > $OUT/run.json 2> $OUT/run.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('$kind mains $mains sides $sides', d['value'])
except Exception as e: print('$kind $mains $sides FAILED', e)" | tee -a $OUT/summary.txt
done < $OUT/cands.txt
sort -k6 -n -r $OUT/summary.txt | head -8
