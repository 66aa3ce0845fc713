#!/bin/bash
# Round 6, call 55: call 54 again with every pool stream bound in a fixed order first (runtime.bind_streams).
OUT=gpurun_out/r06_55; mkdir -p $OUT
python tools/experiments/stream_queue_map.py > $OUT/map.txt 2> $OUT/map.err; cat $OUT/map.txt; tail -2 $OUT/map.err
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3"
summ() {
python - $1 "$2" <<'PY' | tee -a gpurun_out/r06_55/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "step_only", d.get("step_only_fps"), "host", d.get("host_frames_fps"), "plain", (d.get("plain_step_fps") or {}).get("deferred_association"), "multi", (d.get("multi_sequence_fps") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
python bench.py $FAST > $OUT/fast.json 2> $OUT/fast.err; summ $OUT/fast.json "cfg2 all legs"
python bench.py $FAST --no-single-sequence > $OUT/lanes.json 2> $OUT/lanes.err; summ $OUT/lanes.json "cfg2 lanes alone"
for sides in "12,8,14" "13,12,15" "4,6,19" "20,24,28" "3,7,11" "9,10,11"; do
  TF_LANE_SIDES=$sides python bench.py $FAST --no-single-sequence > $OUT/l.json 2> $OUT/l.err; summ $OUT/l.json "cfg2 lanes alone sides $sides"
  TF_LANE_SIDES=$sides python bench.py $FAST > $OUT/l.json 2> $OUT/l.err; summ $OUT/l.json "cfg2 all legs sides $sides"
done
