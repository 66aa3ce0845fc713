# Round 6, call 31: the halo convolution for Cin % 32 and <= 32 output channels (the mask head's lay2 / lay4 / lay5): kernel tests, cfg-5 parity,
# bench cfg5 + kernel statistics, cfg 2 sanity
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_31
mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_models_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -s -k "conv3x3 or upsample_add or one_channel or mask or cfg5 or segm" > $O/pytest_mask.txt 2>&1; tail -4 $O/pytest_mask.txt; grep "^conv " $O/pytest_mask.txt | tail -6
timeout 900 python bench.py --config cfg5 --no-cpu-baseline --no-fp32-exact --no-split3 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python3 -c "
import json
d=json.load(open('$O/bench_cfg5.json'))
print('cfg5', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'plain', d.get('plain_step_fps') and d['plain_step_fps']['deferred_association'], 'multi', d.get('multi_sequence_fps'))"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --no-split3 --no-fp32-exact --no-parity --no-roofline --sequences 1 --no-single-sequence --steps 24 --warmup 4 --min-seconds 3 > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -70 $f > $O/bench_cfg5_kernel_stats_top70.csv
rm -rf $O/prof
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('$O/bench_cfg5_kernel_stats_top70.csv')))
frames=sum(int(r['Calls']) for r in rows if 'ffn_fused_kernel' in r['Name'])/6
print('frames', frames)
for r in rows[:30]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'^void ','',n); n=re.sub(r'\(.*','',n)[:90]
    print('%-92s %6.1f/frame %9.1f us avg %8.1f us/frame'%(n,int(r['Calls'])/frames,float(r['AverageNs'])/1e3,int(r['TotalDurationNs'])/frames/1e3))
PY
timeout 600 python bench.py --no-cpu-baseline --no-fp32-exact --no-split3 --no-roofline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python3 -c "
import json
d=json.load(open('$O/bench_cfg2.json')); print('cfg2', d['value'], d.get('plain_step_fps'), d['parity']['ids_equal'], d['parity']['pipelined']['ids_equal'])"
