#!/bin/bash
# Round 6, call 59: after the last edits of bench.py (stream pool bound at start-up / before RCCL): the rank-launch tests, the schedule
# tests, and the line with the side legs off.
OUT=gpurun_out/r06_59; mkdir -p $OUT
python -m pytest tests/test_bench_ranks_gpu.py tests/test_graph_schedule.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python bench.py --no-cpu-baseline --no-fp32-exact --no-split3 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_59/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'host', d.get('host_frames_fps'), 'plain', d.get('plain_step_fps',{}).get('deferred_association'), 'multi', d.get('multi_sequence_fps',{}).get('value'), 'roofline', d['roofline']['frac'], 'parity', d['parity']['ids_equal'], d['parity']['pipelined']['ids_equal'])
PY
