mkdir -p gpurun_out/r02j
cd $GRAFT_REPO_ROOT
(timeout 300 tools/bin/msda_bench --iters 20 --sets 4 --patterns init,pert,local quad pquad pquad:npass=3,wgs=2 pquad:lds=40 2>&1 | grep -v plan) > gpurun_out/r02j/sweep.log
(timeout 120 tools/bin/msda_bench --iters 10 --sets 1 --fused 1 --trace --patterns pert pquad 2>&1 | grep -v plan) > gpurun_out/r02j/trace.log
(timeout 900 python -m pytest tests/test_msda_gpu.py -x -q -k "tiled or persistent or fused or full_size" 2>&1 | tail -4) > gpurun_out/r02j/pytest_msda.log
(timeout 600 python bench.py --no-cpu-baseline 2>/dev/null) > gpurun_out/r02j/bench.json
