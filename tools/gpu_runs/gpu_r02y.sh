mkdir -p gpurun_out/r02y
cd $GRAFT_REPO_ROOT
(timeout 300 tools/bin/msda_bench --iters 20 --sets 4 --fused 1 --patterns init,pert,local pquad pquad:lds=44 pquad:lds=52 pquad:hy=8,hx=12 pquad:hy=4,hx=8 pquad:lds=52,hy=8,hx=12 2>&1 | grep -v "plan" | grep fused) > gpurun_out/r02y/sweep.log
