#!/bin/bash
# Round 6, call 44: the schedule of the prepared image-only halves -- slots x side streams x frames of look-ahead -- on ONE box:
# cfg 2 and cfg 5 lines (value = pipelined loop with HBM frames, host_frames_fps, plain_step_fps).
OUT=gpurun_out/r06_44; mkdir -p $OUT
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-calibration"
for cfg in cfg2 cfg5; do
for v in "2 1 1" "2 2 1" "4 1 1" "4 2 1" "4 2 2" "4 1 2" "3 1 1" "2 1 1"; do
  set -- $v
  TF_GRAPH_SLOTS=$1 TF_GRAPH_SIDE_STREAMS=$2 TF_GRAPH_LOOKAHEAD=$3 python bench.py --config $cfg $FAST > $OUT/${cfg}_s$1_t$2_l$3.json 2> $OUT/${cfg}_s$1_t$2_l$3.err
  python - $OUT/${cfg}_s$1_t$2_l$3.json "$cfg slots=$1 streams=$2 lookahead=$3" <<'PY' | tee -a $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "step_only", d.get("step_only_fps"), "host", d.get("host_frames_fps"), "plain", (d.get("plain_step_fps") or {}).get("deferred_association"), "multi", (d.get("multi_sequence_fps") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done; done
