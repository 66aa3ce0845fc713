#!/bin/bash
# Round 6, call 46: are the quantised rates of call 44 / 45 (284 / 308 / 331 / 368 frames/s for the same schedule, by the order in
# which the process created its streams) HIP streams sharing a hardware queue?  GPU_MAX_HW_QUEUES (default 4) x the schedules.
OUT=gpurun_out/r06_46; mkdir -p $OUT
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-calibration"
for q in 8 16 2 ""; do
for v in "2 1 1" "2 2 1" "4 1 1" "4 2 1" "4 2 2"; do
  set -- $v
  tag=q${q:-default}_s$1_t$2_l$3
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  TF_GRAPH_SLOTS=$1 TF_GRAPH_SIDE_STREAMS=$2 TF_GRAPH_LOOKAHEAD=$3 python bench.py $FAST > $OUT/cfg2_$tag.json 2> $OUT/cfg2_$tag.err
  python - $OUT/cfg2_$tag.json "cfg2 hw_queues=${q:-default} slots=$1 streams=$2 lookahead=$3" <<'PY' | tee -a $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "step_only", d.get("step_only_fps"), "host", d.get("host_frames_fps"), "plain", (d.get("plain_step_fps") or {}).get("deferred_association"), (d.get("plain_step_fps") or {}).get("association_before_return"), "multi", (d.get("multi_sequence_fps") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done; done
