# Round 3, call 17: the driver's bench command (CPU leg in a fresh process), cfg 5 / cfg 1 lines, the 64-frame tracker test
mkdir -p gpurun_out/r03_17
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03_17
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $*" | tee -a $O/timeline.txt; }
TF_BENCH_WATCHDOG=150 timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
stamp "bench default rc $?"
grep -v amdgpu $O/bench_default.err | tail -5 | cut -c1-200
cut -c1-5000 $O/bench_default.json
timeout 300 python -m pytest tests/test_full_size_gpu.py -m gpu -q -x -k "64_frames" > $O/pytest64.txt 2>&1
tail -4 $O/pytest64.txt
stamp "pytest 64"
TF_BENCH_WATCHDOG=100 timeout 240 python bench.py --config cfg5 --no-cpu-baseline --no-roofline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
stamp "cfg5 rc $?"
grep -v amdgpu $O/bench_cfg5.err | tail -30 | cut -c1-160
timeout 200 python bench.py --config cfg1 --no-cpu-baseline --no-roofline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
python tools/summarize_bench.py $O | tee $O/summary.txt
stamp "done"
