#!/bin/bash
# Round 6, call 51: streams picked by pool index (runtime.pool_stream: sequence on high-priority stream 0, side streams 1 and 5),
# NARROW schedule for interleaved lanes: the full default line (every leg) twice, cfg 4 / cfg 5, the pipelined tests.
OUT=gpurun_out/r06_51; mkdir -p $OUT
python -m pytest tests/test_models_gpu.py -m gpu -x -q -k "prepare or pipelined or graphed or mask or sequences" > $OUT/pytest_models.txt 2>&1; tail -3 $OUT/pytest_models.txt
python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "pipelined or unobserved or multi_frame" > $OUT/pytest_full.txt 2>&1; tail -3 $OUT/pytest_full.txt
summ() {
python - $1 "$2" <<'PY' | tee -a gpurun_out/r06_51/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "step_only", d.get("step_only_fps"), "host", d.get("host_frames_fps"), "plain", (d.get("plain_step_fps") or {}).get("deferred_association"), (d.get("plain_step_fps") or {}).get("association_before_return"), "multi", (d.get("multi_sequence_fps") or {}).get("value"), "six", d.get("split6_fps"), "fp32", d.get("fp32_exact_fps"), d.get("single_sequence_fp32_exact_fps"), "parity", (d.get("parity") or {}).get("ids_equal"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; summ $OUT/bench_default.json "cfg2 full"
python bench.py --no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3 > $OUT/bench_fast.json 2> $OUT/bench_fast.err; summ $OUT/bench_fast.json "cfg2 fast"
python bench.py --no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3 --sequences 2 > $OUT/bench_seq2.json 2> $OUT/bench_seq2.err; summ $OUT/bench_seq2.json "cfg2 fast 2 sequences"
for c in cfg5 cfg4; do
python bench.py --config $c --no-cpu-baseline --no-roofline --no-fp32-exact --no-parity --no-split3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; summ $OUT/bench_$c.json "$c fast"
done
