# Round 6, call 35: the kernel table of a cfg-3 step after the training fold (what is left outside the libraries' GEMMs / convolutions?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_35
mkdir -p $O
timeout 300 python bench.py --config cfg3 --no-cpu-baseline --no-roofline --steps 2 --warmup 2 --min-seconds 0.1 > /dev/null 2>&1   # MIOpen's find results on disk first
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --config cfg3 --no-cpu-baseline --no-roofline --steps 20 --warmup 4 --min-seconds 0.1 > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -90 $f > $O/bench_cfg3_kernel_stats_top90.csv
rm -rf $O/prof
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('$O/bench_cfg3_kernel_stats_top90.csv')))
steps=sum(int(r['Calls']) for r in rows if 'msda_bwd_f32_sorted2' in r['Name'])/6
print('steps', steps)
tot=0
for r in rows[:70]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'^void ','',n)
    n=re.sub(r'at::native::','',n)[:100]
    per=int(r['TotalDurationNs'])/steps/1e6; tot+=per
    print('%-102s %7.1f/step %8.1f us avg %7.2f ms/step'%(n,int(r['Calls'])/steps,float(r['AverageNs'])/1e3,per))
print('sum', tot)
PY
