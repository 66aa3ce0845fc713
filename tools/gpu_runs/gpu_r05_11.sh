# Round 5, call 11: pquad2 with eight-wave workgroups (one pass of 128 pairs, two workgroups per CU, 78 KB of LDS each)
mkdir -p gpurun_out/r05_11
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_11
cd $R
B=$R/tools/bin/msda_bench
timeout 300 $B --iters 24 --sets 4 --fused 1 --patterns pert,init,local pquad pquad:waves=8,npass=1,wgs=2,lds=78 pquad:waves=8,npass=1,wgs=2,lds=70 pquad:waves=8,npass=1,wgs=2,lds=78,hy=8,hx=12 2>&1 | grep "fused pquad" | tee $O/waves8.txt | cut -c1-150
timeout 100 $B --iters 24 --sets 4 --fused 1 --trace --patterns pert pquad:waves=8,npass=1,wgs=2,lds=78 2>&1 | tail -16 | tee $O/trace.txt | cut -c1-150
