# Round 3, call 1 (trimmed merge of gpu_r03_first.sh + the ablation part of gpu_r03_second.sh; ~20 min):
# hardware parity of every opt-in kernel, kernel-level timings against the defaults, bench.py default vs routes,
# phase ablations of the encoder kernel.   gpurun --timeout 1500 -- 'bash tools/gpu_runs/gpu_r03_01.sh'
mkdir -p gpurun_out/r03_01
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/trackformer_amd/lib:$LD_LIBRARY_PATH
O=gpurun_out/r03_01
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $*" | tee -a $O/timeline.txt; }

# 1. parity of the opt-in kernels / routes on the hardware
stamp "pytest optin"
TF_TEST_OPTIN=1 timeout 600 python -m pytest tests/test_msda_gpu.py tests/test_full_size_gpu.py tests/test_linear_split_gpu.py tests/test_fused_gpu.py tests/test_models_gpu.py -m gpu -q -k optin --durations=15 \
    > $O/pytest_optin.txt 2>&1
tail -40 $O/pytest_optin.txt

# 2. kernel timings, default vs opt-in
stamp "msda kernel timings"
{
echo "## cfg4 decoder forward: msda_fwd_f32_buf (default) vs msda_fwd_f32_direct9"
timeout 120 python tools/bench_msda.py --shapes cfg4_decoder --no-backward
timeout 120 python tools/bench_msda.py --shapes cfg4_decoder --no-backward --option direct9=1
echo "## encoder backward: msda_bwd_f32_sorted (default) vs msda_bwd_f32_sorted2"
timeout 200 python tools/bench_msda.py --shapes cfg2_encoder,cfg3_encoder_n2 --no-forward
timeout 200 python tools/bench_msda.py --shapes cfg2_encoder,cfg3_encoder_n2 --no-forward --option bwd_sorted2=1
echo "## decoder query self-attention: default staging vs TF_MHA_BATCH=1"
timeout 100 python tools/bench_mha.py
TF_MHA_BATCH=1 timeout 100 python tools/bench_mha.py
} > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt | tail -60

stamp "linear bufstore"
{
for shape in "22223 256 1024 packed" "22223 1024 256 packed" "22223 256 256" "22223 256 384" "400 256 256" "66800 64 256" "16700 512 128"; do
    echo "## $shape: default epilogue, then TF_LINEAR_BUFSTORE=1"
    timeout 60 tools/bin/linear_bench $shape | grep -E "us per launch"
    TF_LINEAR_BUFSTORE=1 timeout 60 tools/bin/linear_bench $shape | grep -E "us per launch|differ"
    case "$shape" in *packed*) TF_LINEAR_BUFSTORE=2 timeout 60 tools/bin/linear_bench $shape | grep -E "us per launch|differ";; esac
done
echo "## few rows (decoder): default variant 5 vs the deep-prefetch variant 7 vs the weight-stationary variant 6"
for shape in "400 256 256" "400 256 1024" "400 1024 256" "800 288 288"; do
    for v in 5 7 6; do timeout 60 tools/bin/linear_bench $shape $v | grep -E "us per launch"; done
done
} > $O/linear_bufstore.txt 2>&1
cat $O/linear_bufstore.txt

stamp "ffn fused"
{
for args in "22223 1024 3" "22223 1024 2" "22223 1024 1"; do
    echo "## ffn_bench $args"
    timeout 100 tools/bin/ffn_bench $args
done
for ti in 1 2 3; do
    echo "## tf_linear_res_ln_f32 with $ti row tiles per block"
    TF_LINLN_TI=$ti timeout 100 tools/bin/ffn_bench 22223 128 | grep -A1 "tf_linear_res_ln_f32"
done
TF_LINLN_TI=1 timeout 100 tools/bin/ffn_bench 400 128 | grep -A1 "tf_linear_res_ln_f32"
echo "## hidden 288 (cfg 4)"
TF_LINLN_TI=1 timeout 100 tools/bin/ffn_bench 22223 1024 1 288
} > $O/ffn_fused.txt 2>&1
cat $O/ffn_fused.txt

stamp "conv per layer"
timeout 420 python tools/bench_conv.py > $O/conv_per_layer.txt 2>&1
cat $O/conv_per_layer.txt | tail -80

# 3. encoder kernel: phase ablations
stamp "pquad ablations"
{
echo "## full kernel, and with the pipelined LDS gathers"
timeout 200 tools/bin/msda_bench --iters 24 --sets 4 --patterns pert,init,local --fused 1 pquad pquad:pipe=1 pquad:npass=1,wgs=4,lds=39 pquad:pipe=1,lds=48
for lib in tools/bin/ablate/libtf_msda_abl*.so; do
    [ -e "$lib" ] || continue
    echo "## $lib"
    LD_PRELOAD=$lib timeout 100 tools/bin/msda_bench --iters 24 --sets 4 --patterns pert,init --fused 1 pquad 2>&1 | grep -E "fused +pquad|plain +pquad" | cut -c1-110
done
} > $O/pquad_ablations.txt 2>&1
tail -40 $O/pquad_ablations.txt

# 4. frames/s: defaults against route groups
stamp "bench"
timeout 200 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_default.json 2> $O/bench_cfg2_default.err
stamp "bench all optin"
TF_ALL_OPTIN=1 TF_LINEAR_BUFSTORE=2 TF_LINEAR_DEEP=1 TF_MHA_BATCH=1 TF_MSDA_PQUAD="pipe=1" timeout 200 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_all_optin.json 2> $O/bench_cfg2_all_optin.err
stamp "bench ffn+linln+bufstore"
TF_FFN_FUSED=1 TF_LINLN_FUSED=1 TF_LINEAR_BUFSTORE=1 timeout 200 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_ffn_linln_bufstore.json 2> $O/bench_cfg2_ffn_linln_bufstore.err
stamp "bench conv routes"
timeout 200 python bench.py --no-cpu-baseline --no-roofline --conv1x1-split --conv3x3-split > $O/bench_cfg2_conv1x1_3x3.json 2> $O/bench_cfg2_conv1x1_3x3.err
stamp "bench small routes"
TF_STEM_POOL_FUSED=1 TF_POS_ADD_FUSED=1 TF_BOX_REFINE_FUSED=1 TF_MHA_BATCH=1 TF_BIAS_ACT_BATCH=1 TF_HEADS_SPLIT=1 timeout 200 python bench.py --no-cpu-baseline --no-roofline --input-proj-fused > $O/bench_cfg2_small_routes.json 2> $O/bench_cfg2_small_routes.err
stamp "bench cfg5 lazy"
TF_LAZY_MASKS=1 timeout 200 python bench.py --config cfg5 --no-cpu-baseline --no-roofline > $O/bench_cfg5_lazy_masks.json 2> $O/bench_cfg5_lazy_masks.err
stamp "bench cfg4 optin"
TF_MSDA_DIRECT9=1 TF_FFN_FUSED=1 TF_LINLN_FUSED=1 timeout 200 python bench.py --config cfg4 --no-cpu-baseline --no-roofline > $O/bench_cfg4_optin.json 2> $O/bench_cfg4_optin.err
for f in $O/bench_*.err; do echo "== $f"; tail -3 $f; done
python tools/summarize_bench.py $O | tee $O/summary.txt
stamp "done"
