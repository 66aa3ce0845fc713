# Round 6, call 16: cfg 3 after the collector fix (engine.settle_heap), fused AdamW, one matcher pass for all decoder layers
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_16
mkdir -p $O
timeout 300 python tools/experiments/train_step_drift.py --settle 2>/dev/null > $O/drift_settle.txt; cut -c1-120 $O/drift_settle.txt
timeout 600 python tools/train_profile.py --steps 4 2>/dev/null > $O/train_profile.txt; head -14 $O/train_profile.txt
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "train or cfg3 or loss" > $O/pytest_train.txt 2>&1; tail -3 $O/pytest_train.txt
timeout 600 python bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cut -c1-400 $O/bench_cfg3.json
