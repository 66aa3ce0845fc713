# Round 6, calls 32 / 33: the per-kernel table of a cfg-4 frame (multi-frame model, hidden 288) before / after the mask-tag fix
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_33
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --config cfg4 --no-cpu-baseline --no-split3 --no-fp32-exact --no-parity --no-roofline --sequences 1 --no-single-sequence --steps 60 --warmup 8 --min-seconds 3 > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -80 $f > $O/bench_cfg4_kernel_stats_top80.csv
rm -rf $O/prof
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('$O/bench_cfg4_kernel_stats_top80.csv')))
frames=sum(int(r['Calls']) for r in rows if 'ffn_fused_kernel' in r['Name'])/24
print('frames', frames)
tot=0
for r in rows[:60]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'^void ','',n); n=re.sub(r'\(.*','',n)[:90]
    per=int(r['TotalDurationNs'])/frames/1e3; tot+=per
    print('%-92s %6.1f/frame %9.1f us avg %8.1f us/frame'%(n,int(r['Calls'])/frames,float(r['AverageNs'])/1e3,per))
print('sum of the 60', tot)
PY
