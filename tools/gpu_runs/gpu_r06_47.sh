#!/bin/bash
# Round 6, call 47: tools/experiments/stream_queue_map.py -- the pipelined loop with the sequence's stream and the side streams placed
# on chosen hardware queues (pool stream i <-> queue i mod GPU_MAX_HW_QUEUES), default 4 queues and 8.
OUT=gpurun_out/r06_47; mkdir -p $OUT
python tools/experiments/stream_queue_map.py > $OUT/map_q4.txt 2> $OUT/map_q4.err; cat $OUT/map_q4.txt; tail -3 $OUT/map_q4.err
GPU_MAX_HW_QUEUES=8 python tools/experiments/stream_queue_map.py > $OUT/map_q8.txt 2> $OUT/map_q8.err; cat $OUT/map_q8.txt; tail -3 $OUT/map_q8.err
