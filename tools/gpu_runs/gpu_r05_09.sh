# Round 5, call 9: phase trace of pquad2 with ONE workgroup per CU (uncontended phase durations)
mkdir -p gpurun_out/r05_09
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_09
cd $R
B=$R/tools/bin/msda_bench
timeout 100 $B --iters 24 --sets 4 --fused 1 --trace --patterns pert pquad:wgs=1 2>&1 | tail -16 | tee $O/trace_wgs1.txt | cut -c1-150
timeout 100 $B --iters 24 --sets 4 --fused 1 --trace --patterns pert pquad:wgs=2 2>&1 | tail -16 | tee $O/trace_wgs2.txt | cut -c1-150
