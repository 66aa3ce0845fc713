# Round 6, closing call: the whole -m gpu suite + smoke() on the closing tree, the bench line of every BASELINE configuration, and the
# per-kernel table of a cfg-2 frame.  Outputs -> gpurun_out/r06_closing/, copied to profiles/r06_* by hand.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_closing
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu_all.txt 2>&1; tail -4 $O/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 - <<PY
import json
d=json.load(open('$O/bench_default.json'))
print('cfg2 value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'host', d.get('host_frames_fps'), 'plain', d.get('plain_step_fps') and (d['plain_step_fps']['deferred_association'], d['plain_step_fps']['association_before_return']), 'multi', d.get('multi_sequence_fps'), 'six', d.get('split6_fps'), 'fp32', d.get('fp32_exact_fps'), d.get('single_sequence_fp32_exact_fps'))
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['kernel'][:40]); print('parity', d['parity']['ids_equal'], d['parity']['max_abs_boxes'], d['parity']['max_abs_logits'], d['parity']['pipelined']['ids_equal'], d['parity']['pipelined']['frames_prepared']); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['c_port']['value'])
PY
for c in cfg1 cfg3 cfg4 cfg5; do
  timeout 900 python bench.py --config $c --no-fp32-exact --no-split3 > $O/bench_$c.json 2> $O/bench_$c.err
  python3 -c "
import json
d=json.load(open('$O/bench_$c.json'))
print('$c', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'host', d.get('host_frames_fps'), 'plain', d.get('plain_step_fps') and (d['plain_step_fps']['deferred_association'], d['plain_step_fps']['association_before_return']), 'multi', d.get('multi_sequence_fps'), 'roofline', d.get('roofline') and (d['roofline'].get('frac'), d['roofline'].get('avg_launch_us')), 'cpu', d.get('cpu_baseline') and d['cpu_baseline'].get('value'))"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-split3 --no-fp32-exact --no-parity --no-roofline --steps 60 --warmup 8 > $O/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -81 $f > $O/bench_kernel_stats_top80.csv
rm -rf $O/prof_bench
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roof -- python $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/bench_roofline_only.json 2> $O/prof_roof.log
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_roof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 $f > $O/bench_roofline_kernel_stats.csv
rm -rf $O/prof_roof
cut -c1-200 $O/bench_roofline_kernel_stats.csv; cut -c1-400 $O/bench_roofline_only.json
