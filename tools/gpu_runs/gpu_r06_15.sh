# Round 6, call 15: cfg 3: which stage of the step is slow when there are >= 12 track queries?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_15
mkdir -p $O
timeout 600 python tools/train_profile.py --steps 2 --per-step 16 2>/dev/null > $O/per_step.txt; grep "^step" $O/per_step.txt | cut -c1-330
