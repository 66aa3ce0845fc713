# Round 4, call 9: validation of the closing tree (fp16 split product as the default) + the round's evidence on it:
#   1. the whole GPU suite (with durations), smoke()        2. the bench line with every leg, incl. the CPU baseline
#   3. rocprofv3 --kernel-trace --stats of the bench command (per-kernel table of a frame) and of the roofline kernel
#   4. PMC passes (own runs, --kernel-trace only): HBM traffic of the encoder kernel; matrix-core utilisation of the dense kernels
mkdir -p gpurun_out/r04_09
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_09
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -24 | tee $O/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_09/bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps', 'single_sequence_fp32_exact_fps', 'split6_fps', 'split3_fps')})
print(d['parity']); print(d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('traffic')); print(d['cpu_baseline'])
print(d['dtype'][:120]); print(json.dumps((d.get('mfma_utilisation') or {}).get('live'))[:900])
PY
cd /tmp && export TMPDIR=/tmp
B=$R/tools/bin/msda_bench
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_msda -- $B --iters 24 --sets 4 --fused 1 --patterns pert pquad > $O/stats_msda.log 2>&1
f=$(find $O/stats_msda -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/msda_fwd_pquad_kernel_stats.csv
timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --iters 8 --sets 4 --fused 1 --patterns pert pquad > $O/fetch.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --iters 8 --sets 4 --fused 1 --patterns pert pquad > $O/write.log 2>&1
for d in fetch write; do
  f=$(find $O/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/$d.json --match msda_fwd > /dev/null
done
timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/mfma_lin1 -- $R/tools/bin/linear_bench 22223 256 1024 packed > $O/mfma_lin1.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/mfma_lin2 -- $R/tools/bin/linear_bench 22223 256 256 > $O/mfma_lin2.log 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/mfma_ffn -- $R/tools/bin/ffn_bench 22223 1024 > $O/mfma_ffn.log 2>&1
for d in mfma_lin1 mfma_lin2 mfma_ffn; do
  f=$(find $O/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/$d.json > /dev/null
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bench -- python $R/bench.py --no-cpu-baseline --no-parity --no-fp32-exact --no-split3 --no-single-sequence --no-roofline --steps 60 --warmup 8 > $O/stats_bench.log 2>&1
f=$(find $O/stats_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -60 $f > $O/bench_kernel_stats_top60.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/mfma_frame -- python $R/bench.py --no-cpu-baseline --no-parity --no-fp32-exact --no-split3 --no-single-sequence --no-roofline --no-graph --sequences 1 --steps 4 --warmup 2 --min-seconds 0.1 > $O/mfma_frame.log 2>&1
f=$(find $O/mfma_frame -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/mfma_frame.json > /dev/null
rm -rf $O/stats_msda $O/fetch $O/write $O/mfma_lin1 $O/mfma_lin2 $O/mfma_ffn $O/stats_bench $O/mfma_frame
cd $R
head -3 $O/msda_fwd_pquad_kernel_stats.csv | cut -c1-200
head -24 $O/bench_kernel_stats_top60.csv | cut -c1-200
