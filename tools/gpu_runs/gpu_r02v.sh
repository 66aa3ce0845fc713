mkdir -p gpurun_out/r02v
cd $GRAFT_REPO_ROOT
(TF_SPLIT_LINEAR=0 timeout 300 python tools/debug_determinism.py 2>&1 | tail -8) > gpurun_out/r02v/det_nosplit.log
(timeout 300 python tools/debug_determinism.py 2>&1 | tail -8) > gpurun_out/r02v/det_split.log
