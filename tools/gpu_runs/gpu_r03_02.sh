# Round 3, call 2: the MSDeformAttn variants call 1 could not time (tools/bench_msda.py shadowed its own argument),
# the remaining single-route bench runs.
mkdir -p gpurun_out/r03_02
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/trackformer_amd/lib:$LD_LIBRARY_PATH
O=gpurun_out/r03_02
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $*" | tee -a $O/timeline.txt; }
{
echo "## cfg4 decoder forward: msda_fwd_f32_buf (default) vs msda_fwd_f32_direct9"
timeout 120 python tools/bench_msda.py --shapes cfg4_decoder --no-backward
timeout 120 python tools/bench_msda.py --shapes cfg4_decoder --no-backward --option direct9=1
echo "## encoder backward: msda_bwd_f32_sorted (default) vs msda_bwd_f32_sorted2"
timeout 200 python tools/bench_msda.py --shapes cfg2_encoder,cfg3_encoder_n2 --no-forward
timeout 200 python tools/bench_msda.py --shapes cfg2_encoder,cfg3_encoder_n2 --no-forward --option bwd_sorted2=1
} > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
stamp "bench single routes"
b() { name=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --no-roofline > $O/bench_$name.json 2> $O/bench_$name.err; stamp "$name"; }
b cfg2_default TF_NONE=1
b cfg2_heads_split TF_HEADS_SPLIT=1
b cfg2_bias_act_batch TF_BIAS_ACT_BATCH=1
b cfg2_pos_add TF_POS_ADD_FUSED=1
b cfg2_box_refine TF_BOX_REFINE_FUSED=1
b cfg2_deep TF_LINEAR_DEEP=1 TF_LINEAR_BUFSTORE=1
b cfg2_stem TF_STEM_POOL_FUSED=1 TF_STEM_CONV_SPLIT=1
timeout 200 python bench.py --no-cpu-baseline --no-roofline --input-proj-fused > $O/bench_cfg2_input_proj.json 2> $O/bench_cfg2_input_proj.err
TF_ALL_OPTIN=1 TF_LINEAR_BUFSTORE=1 TF_LINEAR_DEEP=1 timeout 200 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_all_optin_b1.json 2> $O/bench_cfg2_all_optin_b1.err
TF_ALL_OPTIN=1 TF_LINEAR_BUFSTORE=1 TF_LINEAR_DEEP=1 TF_CONV_SPLIT_SKIP="2048x512x1x1" timeout 200 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_all_optin_skip.json 2> $O/bench_cfg2_all_optin_skip.err
stamp "cfg4/cfg5"
timeout 200 python bench.py --config cfg4 --no-cpu-baseline --no-roofline > $O/bench_cfg4_default.json 2> $O/bench_cfg4_default.err
TF_MSDA_DIRECT9=1 timeout 200 python bench.py --config cfg4 --no-cpu-baseline --no-roofline > $O/bench_cfg4_direct9.json 2> $O/bench_cfg4_direct9.err
TF_ALL_OPTIN=1 TF_LINEAR_BUFSTORE=1 TF_LINEAR_DEEP=1 TF_MSDA_DIRECT9=1 timeout 200 python bench.py --config cfg4 --no-cpu-baseline --no-roofline > $O/bench_cfg4_all_optin.json 2> $O/bench_cfg4_all_optin.err
TF_ALL_OPTIN=1 TF_LINEAR_BUFSTORE=1 TF_LINEAR_DEEP=1 TF_LAZY_MASKS=1 timeout 200 python bench.py --config cfg5 --no-cpu-baseline --no-roofline > $O/bench_cfg5_all_optin.json 2> $O/bench_cfg5_all_optin.err
TF_ALL_OPTIN=1 TF_LINEAR_BUFSTORE=1 TF_LINEAR_DEEP=1 TF_MSDA_BWD_SORTED2=1 timeout 300 python bench.py --config cfg3 --no-cpu-baseline --no-roofline > $O/bench_cfg3_all_optin.json 2> $O/bench_cfg3_all_optin.err
timeout 300 python bench.py --config cfg3 --no-cpu-baseline --no-roofline > $O/bench_cfg3_default.json 2> $O/bench_cfg3_default.err
for f in $O/bench_*.err; do echo "== $f"; tail -2 $f; done
python tools/summarize_bench.py $O | tee $O/summary.txt
stamp "done"
