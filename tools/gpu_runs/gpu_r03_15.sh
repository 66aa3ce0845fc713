# Round 3, call 15: rocprofv3 evidence -- kernel statistics of the driver's bench command, of the roofline kernel (harness),
# HBM traffic counters of the encoder kernel (one --pmc set per run, --kernel-trace only), matrix-core utilisation of the
# split GEMM / the fused feed-forward block / the 3x3 convolution with the corrected normalisation (tools/pmc_summary.py).
mkdir -p gpurun_out/r03_15
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_15
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
B=$R/tools/bin/msda_bench
# 1. the roofline kernel through the harness: stats + traffic
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_msda -- $B --iters 24 --sets 4 --fused 1 --patterns pert pquad > $O/stats_msda.log 2>&1
f=$(find $O/stats_msda -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/msda_fwd_pquad_kernel_stats.csv
timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --iters 8 --sets 4 --fused 1 --patterns pert pquad > $O/fetch.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --iters 8 --sets 4 --fused 1 --patterns pert pquad > $O/write.log 2>&1
for d in fetch write; do
  f=$(find $O/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/$d.json --match msda_fwd > /dev/null
done
# 2. matrix-core utilisation: SQ_VALU_MFMA_BUSY_CYCLES over the dispatch's own SIMD-cycles
timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/mfma_lin1 -- $R/tools/bin/linear_bench 22223 256 1024 packed > $O/mfma_lin1.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/mfma_lin2 -- $R/tools/bin/linear_bench 22223 256 256 > $O/mfma_lin2.log 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/mfma_ffn -- $R/tools/bin/ffn_bench 22223 1024 3 > $O/mfma_ffn.log 2>&1
for d in mfma_lin1 mfma_lin2 mfma_ffn; do
  f=$(find $O/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/$d.json > /dev/null
done
# 3. the driver's bench command: per-kernel statistics (graph replays are kernels too)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bench -- python $R/bench.py --no-cpu-baseline --no-parity --no-fp32-exact --no-single-sequence --steps 60 --warmup 8 > $O/stats_bench.log 2>&1
f=$(find $O/stats_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 $f > $O/bench_kernel_stats_top45.csv
# 4. matrix-core utilisation inside the model (eager frame): convolutions and GEMMs by kernel family
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/mfma_frame -- python $R/bench.py --no-cpu-baseline --no-parity --no-fp32-exact --no-single-sequence --no-roofline --no-graph --sequences 1 --steps 4 --warmup 2 --min-seconds 0.1 > $O/mfma_frame.log 2>&1
f=$(find $O/mfma_frame -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/mfma_frame.json > /dev/null
rm -rf $O/stats_msda $O/fetch $O/write $O/mfma_lin1 $O/mfma_lin2 $O/mfma_ffn $O/stats_bench $O/mfma_frame
cd $R
cat $O/msda_fwd_pquad_kernel_stats.csv | head -5
python3 - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03_15"
for f in sorted(glob.glob(O+"/*.json")):
    d=json.load(open(f))
    print("==",os.path.basename(f))
    for k,v in d.items():
        keep={a:b for a,b in v.items() if a in ("dispatches","mfma_util","avg_dispatch_us","fetch_bytes_per_dispatch_corrected","write_bytes_per_dispatch")}
        if keep.get("mfma_util",1)>0.005 or "fetch" in f or "write" in f: print("  ",k[:50],keep)
PY
head -12 $O/bench_kernel_stats_top45.csv | cut -c1-200
# 5. backward kernel: tile / halo sweep of msda_bwd_f32_sorted2 at the encoder shapes
{
for tile in "" "8,8" "12,8" "6,16"; do
  for halo in "" "6,10"; do
    echo "## TF_MSDA_BWD_TILE='$tile' TF_MSDA_BWD_HALO='$halo'"
    env ${tile:+TF_MSDA_BWD_TILE=$tile} ${halo:+TF_MSDA_BWD_HALO=$halo} timeout 100 python tools/bench_msda.py --shapes cfg3_encoder_n2 --no-forward --modes local,init 2>&1 | grep -v amdgpu
  done
done
} > $O/bwd_sweep.txt 2>&1
cat $O/bwd_sweep.txt

# 6. the 64-frame tracker fixture, cfg 5 line
cd $R
timeout 300 python -m pytest tests/test_full_size_gpu.py -m gpu -q -x -s -k "64_frame" > $O/pytest64.txt 2>&1
grep -E "64-frame fixture|passed|failed" $O/pytest64.txt | cut -c1-300
TF_BENCH_WATCHDOG=120 timeout 240 python bench.py --config cfg5 --no-cpu-baseline --no-roofline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
echo "cfg5 rc $?"; grep -v amdgpu $O/bench_cfg5.err | tail -12 | cut -c1-160
python tools/summarize_bench.py $O
