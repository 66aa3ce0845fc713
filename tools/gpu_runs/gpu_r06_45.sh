#!/bin/bash
# Round 6, call 45: the frame's result rows written by the post-processing kernel straight into pinned host memory
# (TF_DIRECT_HOST_RESULTS, default on) against device rows + an asynchronous copy, across the schedules of call 44; parity first.
OUT=gpurun_out/r06_45; mkdir -p $OUT
python -m pytest tests/test_models_gpu.py tests/test_fused_gpu.py -m gpu -x -q -k "tracker or postprocess or pipelined" > $OUT/pytest_models.txt 2>&1; tail -3 $OUT/pytest_models.txt
python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "pipelined or unobserved" > $OUT/pytest_full.txt 2>&1; tail -3 $OUT/pytest_full.txt
FAST="--no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-calibration"
for v in "1 2 1 1" "1 2 2 1" "1 4 1 1" "1 4 2 1" "1 4 2 2" "0 2 1 1" "0 2 2 1" "0 4 2 2" "1 2 1 1"; do
  set -- $v
  tag=d$1_s$2_t$3_l$4
  TF_DIRECT_HOST_RESULTS=$1 TF_GRAPH_SLOTS=$2 TF_GRAPH_SIDE_STREAMS=$3 TF_GRAPH_LOOKAHEAD=$4 python bench.py $FAST > $OUT/cfg2_$tag.json 2> $OUT/cfg2_$tag.err
  python - $OUT/cfg2_$tag.json "cfg2 direct=$1 slots=$2 streams=$3 lookahead=$4" <<'PY' | tee -a $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "step_only", d.get("step_only_fps"), "host", d.get("host_frames_fps"), "plain", (d.get("plain_step_fps") or {}).get("deferred_association"), (d.get("plain_step_fps") or {}).get("association_before_return"), "multi", (d.get("multi_sequence_fps") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
