# Round 6, call 23: the whole -m gpu suite + smoke on the tree with every round-6 change, then the per-kernel table of a frame
# (rocprofv3 kernel statistics of the bench command with the side legs off) and the default bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_23
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_all.txt 2>&1; tail -5 $O/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-exact --no-parity --no-roofline --steps 60 --warmup 8 > $O/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -81 $f > $O/bench_kernel_stats_top80.csv
grep -h '"metric"' $O/prof_bench.log | tail -1 > $O/bench_line_under_rocprof.json
rm -rf $O/prof_bench
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 - <<PY
import json
d=json.load(open('$O/bench_default.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'host', d.get('host_frames_fps'), 'multi', d.get('multi_sequence_fps'))
print('roofline', d['roofline']); print('parity', d.get('parity')); print('cpu', d.get('cpu_baseline'))
PY
