# Round 3, call 7: the two-pass encoder kernel compiled for four waves per SIMD (128 VGPRs, 19 spilled): four workgroups
# per CU at 39 KB of LDS against the default three at 52 KB.
mkdir -p gpurun_out/r03_07
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_07
LD_PRELOAD=tools/bin/ablate/libtf_msda_w4.so timeout 300 tools/bin/msda_bench --iters 24 --sets 4 --fused 1 --patterns pert,init,local pquad "pquad:wgs=4,lds=39" "pquad:wgs=4,lds=39,th=6,tw=12" "pquad:wgs=4,lds=39,th=8,tw=8" "pquad:wgs=4,lds=39,hy=4,hx=6" > $O/w4.txt 2>&1
grep -E "fused|plain" $O/w4.txt | cut -c1-130
timeout 100 tools/bin/msda_bench --iters 24 --sets 4 --fused 1 --patterns pert pquad > $O/default.txt 2>&1
grep -E "fused|plain" $O/default.txt | cut -c1-130
