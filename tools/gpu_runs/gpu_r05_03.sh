# Round 5, call 3: the whole GPU suite (no -x) on the tree with the round's new parity tests.
mkdir -p gpurun_out/r05_03
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_03
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1700 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -60 | tee $O/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
