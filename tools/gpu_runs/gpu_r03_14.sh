# Round 3, call 14: the whole GPU suite after the host-side work, the driver's bench command, every config
mkdir -p gpurun_out/r03_14
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03_14
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $*" | tee -a $O/timeline.txt; }
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1
tail -22 $O/pytest_gpu.txt
stamp "pytest done"
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -2 $O/bench_default.err
cut -c1-4000 $O/bench_default.json
stamp "bench default"
timeout 300 python bench.py --config cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-roofline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 300 python bench.py --config cfg1 --no-cpu-baseline --no-roofline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
for f in $O/bench_*.err; do echo "== $f"; tail -2 $f; done
python tools/summarize_bench.py $O | tee $O/summary.txt
stamp "done"
