#!/bin/bash
# Round 6, call 62: under the normal-first binding order: the cfg 2 line, and the lanes of cfg 4 (sequence streams) / cfg 5 (sequence +
# side streams) over a few layouts.
OUT=gpurun_out/r06_62; mkdir -p $OUT
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3"
python bench.py $FAST > $OUT/cfg2.json 2> $OUT/cfg2.err
python -c "
import json
d=json.loads(open('$OUT/cfg2.json').read().strip().splitlines()[-1]); print('cfg2: value', d['value'], 'host', d.get('host_frames_fps'), 'plain', (d.get('plain_step_fps') or {}).get('deferred_association'), 'multi', (d.get('multi_sequence_fps') or {}).get('value'))" | tee -a $OUT/summary.txt
for mains in 4,3,10 8,9,10 5,4,3 0,1,2 4,3,7 2,4,6 0,3,1 6,3,1 5,0,1 1,2,3 0,4,8; do
  TF_LANE_MAINS_NARROW=$mains python bench.py --config cfg4 $FAST --no-single-sequence > $OUT/run.json 2> $OUT/run.err
  python -c "
import json
d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('cfg4 lanes mains $mains', d['value'])" | tee -a $OUT/summary.txt
done
for ms in "5,4,3 1,2,6" "4,3,7 1,14,6" "2,4,6 5,0,14" "0,3,1 11,14,8" "5,4,3 1,14,6" "0,1,2 10,14,13" "4,3,10 1,2,6" "8,9,10 1,2,6"; do
  set -- $ms
  TF_LANE_MAINS=$1 TF_LANE_SIDES_NARROW=$2 python bench.py --config cfg5 $FAST --no-single-sequence > $OUT/run.json 2> $OUT/run.err
  python -c "
import json
d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('cfg5 lanes mains $1 sides $2', d['value'])" | tee -a $OUT/summary.txt
done
python bench.py --config cfg5 $FAST --sequences 1 > $OUT/run.json 2> $OUT/run.err; python -c "
import json
d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('cfg5 single', d['value'], d.get('host_frames_fps'))" | tee -a $OUT/summary.txt
python bench.py --config cfg4 $FAST --sequences 1 > $OUT/run.json 2> $OUT/run.err; python -c "
import json
d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print('cfg4 single', d['value'], d.get('host_frames_fps'))" | tee -a $OUT/summary.txt
