mkdir -p gpurun_out/r02b
cd $GRAFT_REPO_ROOT
export TF_MSDA_VERBOSE=1
(timeout 120 tools/bin/msda_bench --iters 5 --sets 1 --fused 1 --patterns init,local quad pquad pquad:pf=2,wgs=2 pquad:npass=1 pquad:npass=3 pquad:ta=8 2>&1) > gpurun_out/r02b/parity.log
(timeout 300 tools/bin/msda_bench --iters 20 --sets 4 --patterns init,pert quad pquad pquad:pf=0 pquad:pf=2,wgs=2 pquad:npass=1,wgs=4 pquad:npass=1,pf=2,wgs=3 pquad:npass=3,wgs=2 pquad:npass=3,pf=2,wgs=2 pquad:ta=8 pquad:lds=48 pquad:wgs=2 2>&1 | grep -v "pquad plan\|quad plan") > gpurun_out/r02b/sweep.log
(timeout 120 tools/bin/msda_bench --iters 10 --sets 1 --fused 0 --trace --patterns init pquad pquad:pf=2,wgs=2 2>&1) > gpurun_out/r02b/trace.log
