# Round 6, call 14: cfg 3: why does the step time grow with the number of steps (119 -> 182 ms per step after 12 warm-up steps)?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_14
mkdir -p $O
timeout 300 python tools/experiments/train_step_drift.py 2>/dev/null > $O/drift_default.txt; cat $O/drift_default.txt | cut -c1-150
timeout 300 python tools/experiments/train_step_drift.py --lr0 2>/dev/null > $O/drift_lr0.txt; sed -n 1,24p $O/drift_lr0.txt | cut -c1-150
