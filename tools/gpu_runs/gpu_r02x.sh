mkdir -p gpurun_out/r02x
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_msda_gpu.py -x -q -k "head_dim_36 or d36 or tiled or persistent" 2>&1 | tail -4) > gpurun_out/r02x/pytest.log
(TF_MSDA_VERBOSE=1 timeout 300 python bench.py --config cfg4 --roofline-only 2>gpurun_out/r02x/roof.err) > gpurun_out/r02x/roof_cfg4.json
grep -m2 "pquad plan" gpurun_out/r02x/roof.err > gpurun_out/r02x/plan.txt
