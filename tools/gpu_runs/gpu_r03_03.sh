# Round 3, call 3: the whole GPU suite with the new defaults, the driver's bench command, the eager per-frame kernel breakdown
# (launches per frame), cfg 3 / 4 / 5 lines.
mkdir -p gpurun_out/r03_03
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03_03
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $*" | tee -a $O/timeline.txt; }
timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > $O/pytest_gpu.txt 2>&1
tail -25 $O/pytest_gpu.txt
stamp "pytest done"
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
stamp "bench default"
TF_ROUND2_ROUTES=1 TF_LINEAR_BUFSTORE=0 TF_LINEAR_DEEP=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench_round2_routes.json 2> $O/bench_round2_routes.err
timeout 300 python bench.py --no-split-linear --no-cpu-baseline --no-roofline > $O/bench_fp32_exact.json 2> $O/bench_fp32_exact.err
stamp "bench A/B"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof_eager -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 4 --sequences 1 --no-graph --no-roofline > $O/prof_eager.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_eager -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/frame_breakdown.py $f $O/e2e_eager_per_frame.txt
rm -rf $O/prof_eager
stamp "eager breakdown"
timeout 300 python bench.py --config cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-roofline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 300 python bench.py --config cfg1 --no-cpu-baseline --no-roofline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
stamp "cfg4/5/1"
timeout 300 python bench.py --config cfg3 --no-cpu-baseline --no-roofline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
TF_MSDA_BWD_SORTED2=0 timeout 300 python bench.py --config cfg3 --no-cpu-baseline --no-roofline > $O/bench_cfg3_sorted1.json 2> $O/bench_cfg3_sorted1.err
timeout 300 python bench.py --config cfg3 --no-cpu-baseline --no-roofline > $O/bench_cfg3_again.json 2> $O/bench_cfg3_again.err
stamp "cfg3"
for f in $O/bench_*.err; do echo "== $f"; tail -2 $f; done
python tools/summarize_bench.py $O | tee $O/summary.txt
cut -c1-1800 $O/bench_default.json
stamp "done"
