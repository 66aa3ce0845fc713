# Round 3, call 24: staging-row permutation (conflict-free LDS stores) in the split GEMM / convolution kernels: per-layer table
mkdir -p gpurun_out/r03_24
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_24
TF_CONV_KSPLIT_POLICY=768,300,8,32 timeout 200 python tools/bench_conv.py > $O/bench_conv_stage_rows.txt 2>&1
cat $O/bench_conv_stage_rows.txt
timeout 120 python -m pytest tests/test_linear_split_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_linear_split.txt
