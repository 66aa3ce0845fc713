# Round 3, call 30: the default bench line (as the driver runs it) after the host-side pass
mkdir -p gpurun_out/r03_30
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 70 python bench.py > gpurun_out/r03_30/bench_default.json 2> gpurun_out/r03_30/bench_default.err
head -c 1500 gpurun_out/r03_30/bench_default.json
