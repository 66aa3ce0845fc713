#!/bin/bash
# Round 6, call 50: stream_queue_map.py with the absolute pool position of the side streams as a variable (default 4 hardware queues).
OUT=gpurun_out/r06_50; mkdir -p $OUT
python tools/experiments/stream_queue_map.py > $OUT/map_q4.txt 2> $OUT/map_q4.err; cat $OUT/map_q4.txt; tail -3 $OUT/map_q4.err
