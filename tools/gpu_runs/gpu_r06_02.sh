# Round 6, call 2: msda_fwd_f32_pquad2 -- output store policy (plain / nt / sc1 / sc0 sc1: the dirty output lines are written back at
# the end of the kernel otherwise), and tiles of 64 pairs (4 waves x 1 pass) at 4 workgroups per CU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_02
mkdir -p $O
B=tools/bin/msda_bench
timeout 200 $B --iters 24 --sets 4 --fused 1 --patterns pert,init,local --trace-dump $O/trace_raw.csv pquad pquad:st=1 pquad:st=2 pquad:st=3 \
   pquad:npass=1,wgs=4,lds=39 pquad:npass=1,wgs=4,lds=39,st=2 pquad:npass=1,wgs=3,lds=52 pquad:npass=1,wgs=5,lds=31 pquad:waves=8,npass=1,wgs=2,lds=78,st=2 > $O/msda_variants.txt 2>&1
grep -v "^  " $O/msda_variants.txt | grep fused | cut -c1-130
