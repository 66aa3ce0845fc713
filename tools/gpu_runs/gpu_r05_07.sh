# Round 5, call 7: pquad2 -- start-up skew between the workgroups of a CU; workgroups per CU / LDS
mkdir -p gpurun_out/r05_07
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_07
cd $R
B=$R/tools/bin/msda_bench
timeout 300 $B --iters 24 --sets 4 --fused 1 --patterns pert pquad pquad:skew=100 pquad:skew=200 pquad:skew=350 pquad:skew=500 pquad:wgs=2,lds=78 pquad:wgs=2 pquad:wgs=1,lds=150 2>&1 | grep "fused pquad" | tee $O/skew.txt | cut -c1-150
