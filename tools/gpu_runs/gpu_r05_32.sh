cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r32
timeout 300 python tools/stream_trace.py > gpurun_out/r32/stream_trace.txt 2>&1
cut -c1-200 gpurun_out/r32/stream_trace.txt
