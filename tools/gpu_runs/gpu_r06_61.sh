#!/bin/bash
# Round 6, call 61: the ORDER in which runtime.bind_streams touches the pool (TF_BIND_ORDER): single sequence + three lanes (default
# layout and a seeded search of 12).
OUT=gpurun_out/r06_61; mkdir -p $OUT
python - <<'PY' > $OUT/cands.txt
import random
r = random.Random(21)
c = [((4,3,7),(1,14,6)), ((0,1,2),(10,14,13)), ((0,1,2),(12,8,14)), ((0,4,8),(13,12,15))]
for _ in range(9):
    mains = tuple(r.sample(range(8), 3))
    c.append((mains, tuple(r.sample([i for i in range(16) if i not in mains], 3))))
for m, s in c:
    print(",".join(map(str, m)), ",".join(map(str, s)))
PY
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3"
for order in normal_first high_first reverse; do
  export TF_BIND_ORDER=$order
  python bench.py $FAST --sequences 1 > $OUT/run.json 2> $OUT/run.err
  python -c "
import json
d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('$order single: value', d['value'], 'host', d.get('host_frames_fps'), 'plain', (d.get('plain_step_fps') or {}).get('deferred_association'))" | tee -a $OUT/summary.txt
  while read mains sides; do
    TF_LANE_MAINS=$mains TF_LANE_SIDES=$sides python bench.py $FAST --no-single-sequence > $OUT/run.json 2> $OUT/run.err
    python -c "
import json
try:
    d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('$order lanes mains $mains sides $sides', d['value'])
except Exception as e: print('$order $mains $sides FAILED', e)" | tee -a $OUT/summary.txt
  done < $OUT/cands.txt
done
