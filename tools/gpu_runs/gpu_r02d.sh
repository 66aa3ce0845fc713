mkdir -p gpurun_out/r02d
cd $GRAFT_REPO_ROOT
(timeout 300 tools/bin/msda_bench --iters 20 --sets 4 --patterns init,pert,local quad pquad pquad:wide=0 pquad:npass=3,wgs=2 pquad:npass=1,wgs=4 pquad:pf=2,wgs=2 pquad:lds=48 2>&1) > gpurun_out/r02d/sweep.log
(timeout 120 tools/bin/msda_bench --iters 10 --sets 1 --fused 1 --trace --patterns init pquad 2>&1) > gpurun_out/r02d/trace.log
