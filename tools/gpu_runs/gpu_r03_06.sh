# Round 3, call 6: phase traces of the encoder kernel, exact chain vs hinted level-0 windows (pert, fused entry, cold cache)
mkdir -p gpurun_out/r03_06
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_06
timeout 200 tools/bin/msda_bench --iters 10 --sets 4 --fused 1 --trace --patterns pert pquad "pquad:hint=1" > $O/trace.txt 2>&1
cat $O/trace.txt | cut -c1-150
