mkdir -p gpurun_out/r02u
cd $GRAFT_REPO_ROOT
(timeout 300 python tools/debug_determinism.py 2>&1 | tail -12) > gpurun_out/r02u/det.log
(TF_LINEAR_VARIANT=0 timeout 300 python tools/debug_determinism.py 2>&1 | tail -4) > gpurun_out/r02u/det_v0.log
