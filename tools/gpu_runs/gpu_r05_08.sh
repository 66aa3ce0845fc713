# Round 5, call 8: pquad2 with greedy staging rounds (one round when every window fits), 2 barriers per round
mkdir -p gpurun_out/r05_08
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_08
cd $R
B=$R/tools/bin/msda_bench
timeout 300 $B --iters 24 --sets 4 --fused 1 --patterns pert,init,local pquad pquad:v2=0 2>&1 | grep "fused pquad" | tee $O/rounds.txt | cut -c1-150
timeout 100 $B --iters 24 --sets 4 --fused 1 --trace --patterns pert pquad 2>&1 | tail -16 | tee $O/trace.txt | cut -c1-150
