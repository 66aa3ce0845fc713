# Round 3, SECOND GPU call: one bench.py run per opt-in route (what each is worth on its own), cfg 4 / cfg 5, and the phase
# ablations of the encoder kernel.  Same preparation as gpu_r03_first.sh; ~35 min.
#     gpurun --timeout 2700 -- 'bash tools/gpu_runs/gpu_r03_second.sh'
mkdir -p gpurun_out/r03b
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/trackformer_amd/lib:$LD_LIBRARY_PATH
O=gpurun_out/r03b

# 3. frames/s, one opt-in route at a time
timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_default.json 2> $O/bench_cfg2_default.err
TF_LINEAR_BUFSTORE=1 timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_bufstore.json 2> $O/bench_cfg2_bufstore.err
timeout 240 python bench.py --no-cpu-baseline --no-roofline --conv1x1-split > $O/bench_cfg2_conv1x1.json 2> $O/bench_cfg2_conv1x1.err
timeout 240 python bench.py --no-cpu-baseline --no-roofline --input-proj-fused > $O/bench_cfg2_input_proj.json 2> $O/bench_cfg2_input_proj.err
timeout 240 python bench.py --no-cpu-baseline --no-roofline --conv1x1-split --conv3x3-split > $O/bench_cfg2_conv1x1_3x3.json 2> $O/bench_cfg2_conv1x1_3x3.err
TF_LINEAR_BUFSTORE=1 TF_LINEAR_DEEP=1 timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_bufstore_deep.json 2> $O/bench_cfg2_bufstore_deep.err
TF_FFN_FUSED=1 timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_ffn_fused.json 2> $O/bench_cfg2_ffn_fused.err
TF_STEM_POOL_FUSED=1 TF_STEM_CONV_SPLIT=1 timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_stem_conv.json 2> $O/bench_cfg2_stem_conv.err
TF_HEADS_SPLIT=1 timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_heads_split.json 2> $O/bench_cfg2_heads_split.err
TF_POS_ADD_FUSED=1 timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_pos_add.json 2> $O/bench_cfg2_pos_add.err
TF_STEM_POOL_FUSED=1 timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_stem_pool.json 2> $O/bench_cfg2_stem_pool.err
TF_FFN_FUSED=1 TF_LINLN_FUSED=1 timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_ffn_linln.json 2> $O/bench_cfg2_ffn_linln.err
TF_BOX_REFINE_FUSED=1 TF_MHA_BATCH=1 TF_BIAS_ACT_BATCH=1 timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_box_refine.json 2> $O/bench_cfg2_box_refine.err
TF_ALL_OPTIN=1 TF_LINEAR_BUFSTORE=2 TF_LINEAR_DEEP=1 TF_MHA_BATCH=1 TF_MSDA_PQUAD="pipe=1" timeout 240 python bench.py --no-cpu-baseline --no-roofline > $O/bench_cfg2_all_optin.json 2> $O/bench_cfg2_all_optin.err
timeout 240 python bench.py --config cfg4 --no-cpu-baseline --no-roofline > $O/bench_cfg4_default.json 2> $O/bench_cfg4_default.err
TF_MSDA_DIRECT9=1 timeout 240 python bench.py --config cfg4 --no-cpu-baseline --no-roofline > $O/bench_cfg4_direct9.json 2> $O/bench_cfg4_direct9.err
TF_FFN_FUSED=1 TF_LINLN_FUSED=1 timeout 240 python bench.py --config cfg4 --no-cpu-baseline --no-roofline > $O/bench_cfg4_ffn_linln.json 2> $O/bench_cfg4_ffn_linln.err
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-roofline > $O/bench_cfg5_default.json 2> $O/bench_cfg5_default.err
TF_LAZY_MASKS=1 timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-roofline > $O/bench_cfg5_lazy_masks.json 2> $O/bench_cfg5_lazy_masks.err
cat $O/bench_cfg5_default.json $O/bench_cfg5_lazy_masks.json | cut -c1-260
cat $O/bench_cfg2_default.json $O/bench_cfg2_bufstore.json $O/bench_cfg2_bufstore_deep.json $O/bench_cfg2_conv1x1.json $O/bench_cfg2_input_proj.json $O/bench_cfg2_conv1x1_3x3.json $O/bench_cfg2_ffn_fused.json $O/bench_cfg2_ffn_linln.json $O/bench_cfg2_stem_pool.json $O/bench_cfg2_stem_conv.json $O/bench_cfg2_pos_add.json $O/bench_cfg2_box_refine.json $O/bench_cfg2_all_optin.json $O/bench_cfg4_default.json $O/bench_cfg4_direct9.json $O/bench_cfg4_ffn_linln.json | cut -c1-260

# 4. where the encoder kernel's time goes: the kernel without one phase at a time (results wrong by design)
{
echo "## full kernel, and with the pipelined LDS gathers (pquad_pipe: bit-identical, rolling reads in flight)"
timeout 240 tools/bin/msda_bench --iters 24 --sets 4 --patterns pert,init,local --fused 1 pquad pquad:pipe=1 pquad:npass=1,wgs=4,lds=39 pquad:pipe=1,lds=48
for lib in tools/bin/ablate/libtf_msda_abl*.so; do
    [ -e "$lib" ] || continue
    echo "## $lib"
    LD_PRELOAD=$lib timeout 120 tools/bin/msda_bench --iters 24 --sets 4 --patterns pert,init --fused 1 pquad 2>&1 | grep -E "fused +pquad|plain +pquad" | cut -c1-110
done
} > $O/pquad_ablations.txt 2>&1
tail -30 $O/pquad_ablations.txt

# one table of every bench line of this call
python tools/summarize_bench.py $O | tee $O/summary.txt
