# Round 4, call 22: the cfg-5 tracker test with masks for every query (mask head inside the detector call), mask-head route on.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_22
timeout 40 python -m pytest tests/test_full_size_gpu.py -m gpu -q -x -k "test_cfg5_tracker_with_masks_800x1333_matches_reference and masks_for_every_query" 2>&1 | tail -2 | tee gpurun_out/r04_22/pytest_cfg5_tracker_all.txt
