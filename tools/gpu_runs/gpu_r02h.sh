mkdir -p gpurun_out/r02h
cd $GRAFT_REPO_ROOT
(timeout 300 tools/bin/msda_bench --iters 20 --sets 4 --patterns init,pert quad quad:npass=1,lds=26 quad:npass=1,lds=24 quad:npass=1,lds=20 quad:npass=1,lds=31 quad:npass=2,lds=31 quad:npass=2,lds=26 quad:npass=1,lds=26,th=6,tw=8 quad:npass=1,lds=26,th=4,tw=12 pquad 2>&1 | grep -v plan) > gpurun_out/r02h/sweep.log
