# Round 6, calls 29 / 38: where do the 14.5 (call 29) / 10.0 ms (call 38) of a cfg-5 frame go?  rocprofv3 kernel statistics of bench.py --config cfg5 (side legs off)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_40
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --no-split3 --no-fp32-exact --no-parity --no-roofline --sequences 1 --no-single-sequence --steps 24 --warmup 4 --min-seconds 4 > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -70 $f > $O/bench_cfg5_kernel_stats_top70.csv
grep -h '"metric"' $O/prof.log | tail -1 > $O/bench_cfg5_line_under_rocprof.json
rm -rf $O/prof
python3 - <<PY
import csv,re,json
d=json.load(open('$O/bench_cfg5_line_under_rocprof.json')); print('value', d['value'], 'steps_timed', d['steps_timed'])
rows=list(csv.DictReader(open('$O/bench_cfg5_kernel_stats_top70.csv')))
frames=sum(int(r['Calls']) for r in rows if 'ffn_fused_kernel' in r['Name'])/6; print('frames', frames)
for r in rows[:45]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'^void ','',n); n=re.sub(r'\(.*','',n)[:95]
    print('%-97s %6.1f/frame %9.1f us avg %8.1f us/frame'%(n,int(r['Calls'])/frames,float(r['AverageNs'])/1e3,int(r['TotalDurationNs'])/frames/1e3))
PY
