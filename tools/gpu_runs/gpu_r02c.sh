mkdir -p gpurun_out/r02c
cd $GRAFT_REPO_ROOT
(timeout 300 tools/bin/msda_bench --iters 20 --sets 4 --patterns init,pert,local quad pquad pquad:wide=0 pquad:skew=150 pquad:skew=250 pquad:skew=350 pquad:skew=500 pquad:pf=2,wgs=2 pquad:pf=2,wgs=2,skew=300 pquad:npass=1,wgs=4 pquad:npass=1,wgs=4,skew=150 pquad:npass=3,wgs=2 pquad:npass=3,wgs=2,skew=400 2>&1) > gpurun_out/r02c/sweep.log
(timeout 120 tools/bin/msda_bench --iters 10 --sets 1 --fused 0 --trace --patterns init pquad pquad:skew=250 2>&1) > gpurun_out/r02c/trace.log
