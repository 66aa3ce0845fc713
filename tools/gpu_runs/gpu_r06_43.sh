#!/bin/bash
# Round 6, call 43: two frames of look-ahead (GraphedDetector: 4 slots, 2 side streams; Tracker.look_ahead; the bench loop).
# Parity of the pipelined loops first, then the bench line A/B (--look-ahead 1 = the previous schedule).
set -x
OUT=gpurun_out/r06_43; mkdir -p $OUT
python -m pytest tests/test_models_gpu.py -m gpu -x -q -k "prepare or pipelined or graphed or mask" > $OUT/pytest_models.txt 2>&1; tail -5 $OUT/pytest_models.txt
python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "pipelined or unobserved or multi_frame" > $OUT/pytest_full.txt 2>&1; tail -5 $OUT/pytest_full.txt
python bench.py --look-ahead 1 > $OUT/bench_la1.json 2> $OUT/bench_la1.err; tail -c 1500 $OUT/bench_la1.json
python bench.py > $OUT/bench_la2.json 2> $OUT/bench_la2.err; tail -c 3000 $OUT/bench_la2.json
python bench.py --config cfg5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; tail -c 800 $OUT/bench_cfg5.json
python bench.py --config cfg5 --look-ahead 1 > $OUT/bench_cfg5_la1.json 2> $OUT/bench_cfg5_la1.err; tail -c 800 $OUT/bench_cfg5_la1.json
