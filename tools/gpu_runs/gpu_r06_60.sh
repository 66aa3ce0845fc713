#!/bin/bash
# Round 6, call 60: deeper look-ahead / a third side stream / other side-stream members for cfg 2's single sequence.
OUT=gpurun_out/r06_60; mkdir -p $OUT
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3 --sequences 1"
run() {
  tag=$1; la=$2; shift 2
  env "$@" python bench.py $FAST --look-ahead $la > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json "$tag $*" <<'PY' | tee -a gpurun_out/r06_60/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "host", d.get("host_frames_fps"), "plain", (d.get("plain_step_fps") or {}).get("deferred_association"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run default 2 A=1
run la3_s5_t3 3 TF_GRAPH_LOOKAHEAD=3 TF_GRAPH_SLOTS=5 TF_GRAPH_SIDE_STREAMS=3
run la3_s6_t3 3 TF_GRAPH_LOOKAHEAD=3 TF_GRAPH_SLOTS=6 TF_GRAPH_SIDE_STREAMS=3
run la3_s6_t2 3 TF_GRAPH_LOOKAHEAD=3 TF_GRAPH_SLOTS=6 TF_GRAPH_SIDE_STREAMS=2
run la2_s6_t2 2 TF_GRAPH_SLOTS=6
run la2_s4_t2_first0 2 TF_GRAPH_SIDE_FIRST=0
run la2_s4_t2_spacing8 2 TF_GRAPH_SIDE_SPACING=8
run la2_s4_t2_spacing1 2 TF_GRAPH_SIDE_SPACING=1
run la2_s4_t2_first1_sp2 2 TF_GRAPH_SIDE_SPACING=2
run seq_index2 2 TF_SEQ_STREAM_INDEX=2
run la2_s4_t4 2 TF_GRAPH_SIDE_STREAMS=4
