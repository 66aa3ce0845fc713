cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r22
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r22/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r22/smoke.txt 2>&1
tail -5 gpurun_out/r22/pytest_gpu_all.txt; tail -2 gpurun_out/r22/smoke.txt
