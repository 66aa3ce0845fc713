# Round 4, call 15: phase trace of the stream GEMM's K-slice (tools/stream_trace.py on the -DTF_STREAM_TRACE build).
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r04_15
timeout 300 python tools/stream_trace.py > gpurun_out/r04_15/stream_trace.txt 2>&1
cat gpurun_out/r04_15/stream_trace.txt | cut -c1-170
