#!/bin/bash
# Round 6, call 49: which switch of the new schedule costs cfg 5 (mask head) its 10 %, and the interleaved sequences their rate.
OUT=gpurun_out/r06_49; mkdir -p $OUT
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3"
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
run old_prio_high cfg5 TF_GRAPH_SLOTS=2 TF_GRAPH_SIDE_STREAMS=1 TF_GRAPH_LOOKAHEAD=1 TF_SEQ_STREAM_PRIORITY=-1
run new_prio0 cfg5 TF_SEQ_STREAM_PRIORITY=0
run new_la1 cfg5 TF_GRAPH_LOOKAHEAD=1
run s4_t1_la1_prio0 cfg5 TF_GRAPH_SLOTS=4 TF_GRAPH_SIDE_STREAMS=1 TF_GRAPH_LOOKAHEAD=1 TF_SEQ_STREAM_PRIORITY=0
run s2_t2_la1_prio0 cfg5 TF_GRAPH_SLOTS=2 TF_GRAPH_SIDE_STREAMS=2 TF_GRAPH_LOOKAHEAD=1 TF_SEQ_STREAM_PRIORITY=0
run s3_t1_la2_prio0 cfg5 TF_GRAPH_SLOTS=3 TF_GRAPH_SIDE_STREAMS=1 TF_GRAPH_LOOKAHEAD=2 TF_SEQ_STREAM_PRIORITY=0
run old cfg5 TF_GRAPH_SLOTS=2 TF_GRAPH_SIDE_STREAMS=1 TF_GRAPH_LOOKAHEAD=1 TF_SEQ_STREAM_PRIORITY=0
