#!/bin/bash
# Round 6, call 48: the bench legs with a high-priority sequence stream + two side streams sharing a hardware queue + two frames of
# look-ahead (the new defaults) against the round's previous schedule and the single switches; cfg 2, cfg 5, cfg 4.
OUT=gpurun_out/r06_48; mkdir -p $OUT
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3"
OLD="TF_GRAPH_SLOTS=2 TF_GRAPH_SIDE_STREAMS=1 TF_GRAPH_LOOKAHEAD=1 TF_SEQ_STREAM_PRIORITY=0"
run() {  # tag, config, env...
  tag=$1; cfg=$2; shift 2
  env "$@" python bench.py --config $cfg $FAST > $OUT/${cfg}_$tag.json 2> $OUT/${cfg}_$tag.err
  python - $OUT/${cfg}_$tag.json "$cfg $tag" <<'PY' | tee -a $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "step_only", d.get("step_only_fps"), "host", d.get("host_frames_fps"), "plain", (d.get("plain_step_fps") or {}).get("deferred_association"), (d.get("plain_step_fps") or {}).get("association_before_return"), "multi", (d.get("multi_sequence_fps") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run new cfg2 A=1
run old cfg2 $OLD
run prio0 cfg2 TF_SEQ_STREAM_PRIORITY=0
run spacing1 cfg2 TF_GRAPH_SIDE_SPACING=1
run la1 cfg2 TF_GRAPH_LOOKAHEAD=1
run new2 cfg2 A=1
run new cfg5 A=1
run old cfg5 $OLD
run new cfg4 A=1
run old cfg4 $OLD
run new cfg1 A=1
