# Round 3, call 16: why did the default bench command not finish in call 14?  (stack dumps every 60 s)
mkdir -p gpurun_out/r03_16
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_16
T0=$(date +%s)
TF_BENCH_WATCHDOG=60 timeout 170 python bench.py --no-roofline --no-parity --no-fp32-exact --no-single-sequence --steps 30 --min-seconds 0.5 > $O/bench.json 2> $O/bench.err
echo "rc $? after $(( $(date +%s) - T0 )) s"
grep -v "amdgpu.ids" $O/bench.err | head -80 | cut -c1-200
cut -c1-600 $O/bench.json
