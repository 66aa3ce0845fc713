# Round 6, call 29: where do the 14.5 ms of a cfg-5 frame go?  rocprofv3 kernel statistics of bench.py --config cfg5 (side legs off)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_29
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --no-split3 --no-fp32-exact --no-parity --no-roofline --sequences 1 --steps 24 --warmup 4 --min-seconds 1 > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -70 $f > $O/bench_cfg5_kernel_stats_top70.csv
grep -h '"metric"' $O/prof.log | tail -1 > $O/bench_cfg5_line_under_rocprof.json
rm -rf $O/prof
python3 - <<PY
import csv,re,json
d=json.load(open('$O/bench_cfg5_line_under_rocprof.json')); print('value', d['value'], 'steps_timed', d['steps_timed'])
rows=list(csv.DictReader(open('$O/bench_cfg5_kernel_stats_top70.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:45]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'^void ','',n); n=re.sub(r'\(.*','',n)[:95]
    print('%-97s %7s calls %9.1f us avg %6.2f%%'%(n,r['Calls'],float(r['AverageNs'])/1e3,100*int(r['TotalDurationNs'])/tot))
PY
