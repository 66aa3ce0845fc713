# Round 6, call 1: (a) the new parity tests of the TIMED path (pipelined tracker at 800x1333, graph/audit interplay, prepare aliasing);
# (b) msda_fwd_f32_pquad2: raw per-workgroup phase stamps (by head / XCD offline), head-mixing and static-priority variants;
# (c) one default bench line (step_only_fps / host_frames_fps / parity.path / roofline.kernel from the library)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_01
mkdir -p $O
B=tools/bin/msda_bench
rm -f $O/trace_raw.csv
timeout 120 $B --iters 24 --sets 4 --fused 1 --patterns pert,init --trace-dump $O/trace_raw.csv pquad pquad:hm=1 pquad:hm=2 pquad:prio=1 pquad:prio=2 pquad:hm=2,prio=1 pquad:hm=2,prio=2 pquad:hm=1,prio=2 > $O/msda_variants.txt 2>&1
grep -v "^  " $O/msda_variants.txt | cut -c1-130
timeout 900 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "range_audit or prepare_twice or pipelined" > $O/pytest_new_small.txt 2>&1; tail -3 $O/pytest_new_small.txt
timeout 1200 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "pipelined_tracker_64" > $O/pytest_new_full.txt 2>&1; tail -3 $O/pytest_new_full.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
