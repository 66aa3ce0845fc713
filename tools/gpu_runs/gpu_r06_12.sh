# Round 6, call 12: cfg 3 (training step, bs 2, 800x1333): stage timings, host profile, kernel trace of one steady-state step
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_12
mkdir -p $O
timeout 600 python tools/train_profile.py --steps 3 --cprofile > $O/train_profile.txt 2>&1
head -60 $O/train_profile.txt | cut -c1-180
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 3 --warmup 2 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/train_breakdown.py $f $O/train_step_kernels.txt | head -50 | cut -c1-170
rm -rf $O/prof
tail -2 $O/prof.log | cut -c1-400
