# Round 4, call 21: the cfg-5 tracker test (lazy masks) with the mask-head route on -- what is left of the GPU budget.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_21
timeout 30 python -m pytest tests/test_full_size_gpu.py -m gpu -q -x -k "test_cfg5_tracker_with_masks_800x1333_matches_reference and lazy_masks" 2>&1 | tail -2 | tee gpurun_out/r04_21/pytest_cfg5_tracker.txt
