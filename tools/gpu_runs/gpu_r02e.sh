mkdir -p gpurun_out/r02e
cd $GRAFT_REPO_ROOT
(timeout 120 tools/bin/msda_bench --iters 10 --sets 1 --fused 0 --trace --patterns init pquad pquad:wgs=1 2>&1) > gpurun_out/r02e/trace.log
