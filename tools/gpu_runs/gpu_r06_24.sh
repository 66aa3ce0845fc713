# Round 6, call 24: the backbone of a training step through the inference kernels where nothing is differentiated and with the BN
# fold + one epilogue pass where it is (backbone.set_train_fold): stage times with / without, parity tests, cfg-3 line;
# then the whole -m gpu suite (no -x) and the per-kernel table of a frame (rocprofv3 kernel statistics, side legs off)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_24
mkdir -p $O
timeout 600 python tools/train_profile.py --steps 4 2>/dev/null > $O/train_profile_fold.txt; head -14 $O/train_profile_fold.txt
TF_TRAIN_FOLD=0 timeout 600 python tools/train_profile.py --steps 4 2>/dev/null > $O/train_profile_nofold.txt; head -14 $O/train_profile_nofold.txt
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "train or cfg3 or loss" > $O/pytest_train.txt 2>&1; tail -5 $O/pytest_train.txt
timeout 600 python bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cut -c1-300 $O/bench_cfg3.json
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu_all.txt 2>&1; tail -8 $O/pytest_gpu_all.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-split3 --no-fp32-exact --no-parity --no-roofline --steps 60 --warmup 8 > $O/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -81 $f > $O/bench_kernel_stats_top80.csv
grep -h '"metric"' $O/prof_bench.log | tail -1 > $O/bench_line_under_rocprof.json
rm -rf $O/prof_bench
timeout 300 python bench.py --roofline-only > $O/bench_roofline_only.json 2> $O/bench_roofline_only.err; cut -c1-900 $O/bench_roofline_only.json
