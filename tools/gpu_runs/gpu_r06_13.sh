# Round 6, call 13: cfg 3: is the 2-step warm-up enough?  steady-state kernel trace of one step
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_13
mkdir -p $O
for w in 2 6 12; do timeout 300 python tools/bench_train.py --steps 6 --warmup $w 2>/dev/null | cut -c1-60,230-400; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 3 --warmup 6 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
ls -la $f
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# steps: find optimizer kernels? use the last third of the trace as one step window by time: total trace / (6+3) steps
t0=int(rows[0]['Start_Timestamp']); t1=int(rows[-1]['End_Timestamp'])
# locate step boundaries by the multi_tensor_apply (AdamW) kernel bursts
ad=[i for i,r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name'] or 'adam' in r['Kernel_Name'].lower()]
ends=[]
for k,i in enumerate(ad):
    if k+1==len(ad) or int(rows[ad[k+1]]['Start_Timestamp'])-int(rows[i]['Start_Timestamp'])>20_000_000: ends.append(i)
print('adam bursts',len(ends))
a,b=ends[-2]+1,ends[-1]+1
seg=rows[a:b]
wall=(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e6
agg=collections.defaultdict(lambda:[0,0])
for r in seg:
    n=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','')[:100]
    agg[n][0]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); agg[n][1]+=1
busy=sum(v[0] for v in agg.values())/1e6
out=["# one steady-state cfg-3 training step (bs 2, 800x1333, fp32) under rocprofv3: wall %.1f ms, GPU busy %.1f ms, %d kernel launches"%(wall,busy,len(seg))]
for k,(d,c) in sorted(agg.items(),key=lambda x:-x[1][0])[:45]:
    out.append('%8.2f ms %5d calls  %s'%(d/1e6,c,k))
open("$O/train_step_kernels.txt","w").write("\n".join(out)+"\n")
print("\n".join(out[:50]))
PY
rm -rf $O/prof
