# Round 3, FIRST GPU call (prepared at the end of round 2, when no GPU minutes were left): validates on hardware what
# was written against the SIMT emulator and times it against the defaults.  Before calling:
#     python -m trackformer_amd.build && python tools/build_ablations.py 1 2 4 8 16 32 7
#     gpurun --timeout 2400 -- 'bash tools/gpu_runs/gpu_r03_first.sh'
# Everything lands in gpurun_out/r03a/.  Nothing here changes a default: each result decides one.
mkdir -p gpurun_out/r03a
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/trackformer_amd/lib:$LD_LIBRARY_PATH
O=gpurun_out/r03a

# 1. parity of the opt-in kernels / routes (direct9, bwd_sorted2, conv1x1 split) on the hardware
TF_TEST_OPTIN=1 timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_full_size_gpu.py tests/test_linear_split_gpu.py tests/test_fused_gpu.py tests/test_models_gpu.py -m gpu -q -s -k optin \
    > $O/pytest_optin.txt 2>&1
tail -3 $O/pytest_optin.txt

# 2. kernel timings, default vs opt-in (HIP graph of 50 launches, HIP events)
{
echo "## cfg4 decoder forward: msda_fwd_f32_buf (default) vs msda_fwd_f32_direct9"
timeout 200 python tools/bench_msda.py --shapes cfg4_decoder --no-backward
timeout 200 python tools/bench_msda.py --shapes cfg4_decoder --no-backward --option direct9=1
echo "## encoder backward: msda_bwd_f32_sorted (default) vs msda_bwd_f32_sorted2"
timeout 300 python tools/bench_msda.py --shapes cfg2_encoder,cfg3_encoder_n2 --no-forward
timeout 300 python tools/bench_msda.py --shapes cfg2_encoder,cfg3_encoder_n2 --no-forward --option bwd_sorted2=1
echo "## decoder query self-attention: default staging vs TF_MHA_BATCH=1 (eight loads in flight per thread)"
timeout 120 python tools/bench_mha.py
TF_MHA_BATCH=1 timeout 120 python tools/bench_mha.py
} > $O/kernel_times.txt 2>&1

# 2b. the split GEMMs with the buffer-store epilogue (no vmcnt(0) between stores) against the default epilogue
{
for shape in "22223 256 1024 packed" "22223 1024 256 packed" "22223 256 256" "22223 256 384" "400 256 256" "66800 64 256" "16700 512 128"; do
    echo "## $shape: default epilogue, then TF_LINEAR_BUFSTORE=1"
    timeout 60 tools/bin/linear_bench $shape | grep -E "us per launch"
    TF_LINEAR_BUFSTORE=1 timeout 60 tools/bin/linear_bench $shape | grep -E "us per launch|differ"
    case "$shape" in *packed*) TF_LINEAR_BUFSTORE=2 timeout 60 tools/bin/linear_bench $shape | grep -E "us per launch|differ";; esac
done
echo "## few rows (decoder): default variant 5 vs the deep-prefetch variant 7 vs the weight-stationary variant 6"
for shape in "400 256 256" "400 256 384" "400 256 1024" "400 1024 256" "800 288 288"; do
    for v in 5 7 6; do timeout 60 tools/bin/linear_bench $shape $v | grep -E "us per launch"; done
done
} > $O/linear_bufstore.txt 2>&1
cat $O/linear_bufstore.txt

# 2c. the feed-forward block in one launch (tf_ffn_fused_f32) against linear1 + ReLU, linear2, residual + LayerNorm
{
for args in "22223 1024 3" "22223 1024 2" "22223 1024 1" "5600 1024 3"; do
    echo "## ffn_bench $args"
    timeout 120 tools/bin/ffn_bench $args
done
for ti in 1 2 3; do
    echo "## tf_linear_res_ln_f32 with $ti row tiles per block"
    TF_LINLN_TI=$ti timeout 120 tools/bin/ffn_bench 22223 128 | grep -A1 "tf_linear_res_ln_f32"
done
TF_LINLN_TI=1 timeout 120 tools/bin/ffn_bench 400 128 | grep -A1 "tf_linear_res_ln_f32"
for ti in 1 2; do
    echo "## hidden 288 (cfg 4), $ti row tiles per block"
    TF_LINLN_TI=$ti timeout 120 tools/bin/ffn_bench 22223 1024 $ti 288
done
} > $O/ffn_fused.txt 2>&1
cat $O/ffn_fused.txt

# 2d. the backbone's bottleneck convolutions shape by shape: library convolution + bias_act against the split-product routes
timeout 600 python tools/bench_conv.py > $O/conv_per_layer.txt 2>&1
TF_CONV_SPLITK=0 timeout 600 python tools/bench_conv.py > $O/conv_per_layer_no_splitk.txt 2>&1   # layer3 / layer4 3x3 rows without the split-K launch
cat $O/conv_per_layer.txt

# 3. frames/s: the defaults against every opt-in route at once (the per-option runs are in gpu_r03_second.sh)
timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_default.json 2> $O/bench_cfg2_default.err
TF_ALL_OPTIN=1 TF_LINEAR_BUFSTORE=2 TF_LINEAR_DEEP=1 TF_MHA_BATCH=1 TF_MSDA_PQUAD="pipe=1" timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_all_optin.json 2> $O/bench_cfg2_all_optin.err
TF_FFN_FUSED=1 TF_LINLN_FUSED=1 TF_LINEAR_BUFSTORE=1 timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_ffn_linln_bufstore.json 2> $O/bench_cfg2_ffn_linln_bufstore.err
timeout 240 python bench.py --no-cpu-baseline --no-roofline --conv1x1-split --conv3x3-split > $O/bench_cfg2_conv1x1_3x3.json 2> $O/bench_cfg2_conv1x1_3x3.err
cat $O/bench_cfg2_default.json $O/bench_cfg2_all_optin.json $O/bench_cfg2_ffn_linln_bufstore.json $O/bench_cfg2_conv1x1_3x3.json | cut -c1-260

# one table of every bench line of this call
python tools/summarize_bench.py $O | tee $O/summary.txt
