# phase trace of the activation-stationary GEMM
mkdir -p gpurun_out/r03h
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/trackformer_amd/lib:$LD_LIBRARY_PATH
export LINEAR_BENCH_TRACE=1
{
timeout 60 tools/bin/linear_bench 22223 256 1024 packeda3
timeout 60 tools/bin/linear_bench 22223 256 256 packeda3
} > gpurun_out/r03h/astat_trace.txt 2>&1
