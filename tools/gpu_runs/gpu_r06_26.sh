# Round 6, call 26: why is bench.py --config cfg3 slower with the training fold (129.7 ms) when tools/train_profile.py is faster (88.5)?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_26
mkdir -p $O
timeout 600 python tools/train_profile.py --steps 6 --per-step 6 2>/dev/null > $O/train_profile_fold.txt; head -30 $O/train_profile_fold.txt | cut -c1-250
TF_TRAIN_PUBLISH_BARRIER=1 timeout 600 python bench.py --config cfg3 --no-cpu-baseline --no-roofline > $O/bench_cfg3_barrier.json 2> $O/bench_cfg3_barrier.err
python3 -c "
import json
d=json.load(open('$O/bench_cfg3_barrier.json')); print('cfg3 fold=1 with barriers', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --config cfg3 --no-cpu-baseline --no-roofline --steps 12 --warmup 6 > $O/bench_cfg3_w6.json 2> $O/bench_cfg3_w6.err
python3 -c "
import json
d=json.load(open('$O/bench_cfg3_w6.json')); print('cfg3 fold=1 warmup 6', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --config cfg3 --no-cpu-baseline --no-roofline --min-seconds 0.5 > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -60 $f | cut -c1-260 > $O/bench_cfg3_kernel_stats_top60.csv
rm -rf $O/prof
head -30 $O/bench_cfg3_kernel_stats_top60.csv | cut -c1-200
