# Round 4, call 5: checkpoint validation + the round's rocprofv3 evidence on the current tree.
#   1. the whole GPU suite, smoke()                      2. the bench line (all legs)
#   3. rocprofv3 --kernel-trace --stats of the bench command (per-kernel table) and of the roofline kernel (harness)
#   4. PMC passes (own runs, --kernel-trace only): HBM traffic of the encoder kernel, matrix-core utilisation of the dense
#      kernels in the harnesses and inside an eager frame
mkdir -p gpurun_out/r04_05
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_05
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $O/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_05/bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps', 'single_sequence_fp32_exact_fps', 'split3_fps')}, d['parity'])
print(d['roofline']['avg_launch_us'], d['roofline']['frac'])
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
timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/mfma_lin1 -- $R/tools/bin/linear_bench 22223 256 1024 packed > $O/mfma_lin1.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/mfma_lin2 -- $R/tools/bin/linear_bench 22223 256 256 packed > $O/mfma_lin2.log 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/mfma_ffn -- $R/tools/bin/ffn_bench 22223 1024 > $O/mfma_ffn.log 2>&1
for d in mfma_lin1 mfma_lin2 mfma_ffn; do
  f=$(find $O/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/$d.json > /dev/null
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bench -- python $R/bench.py --no-cpu-baseline --no-parity --no-fp32-exact --no-split3 --no-single-sequence --steps 60 --warmup 8 > $O/stats_bench.log 2>&1
f=$(find $O/stats_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -50 $f > $O/bench_kernel_stats_top50.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/mfma_frame -- python $R/bench.py --no-cpu-baseline --no-parity --no-fp32-exact --no-split3 --no-single-sequence --no-roofline --no-graph --sequences 1 --steps 4 --warmup 2 --min-seconds 0.1 > $O/mfma_frame.log 2>&1
f=$(find $O/mfma_frame -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/mfma_frame.json > /dev/null
rm -rf $O/stats_msda $O/fetch $O/write $O/mfma_lin1 $O/mfma_lin2 $O/mfma_ffn $O/stats_bench $O/mfma_frame
cd $R
head -4 $O/msda_fwd_pquad_kernel_stats.csv | cut -c1-200
python3 - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r04_05"
for f in sorted(glob.glob(O+"/*.json")):
    if "bench_default" in f: continue
    d=json.load(open(f))
    print("==",os.path.basename(f))
    for k,v in d.items():
        if not isinstance(v, dict): continue
        keep={a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("dispatches","mfma_util","avg_dispatch_us","fetch_bytes_per_dispatch_corrected","write_bytes_per_dispatch")}
        if keep.get("mfma_util",1)>0.005 or "fetch" in f or "write" in f: print("  ",k[:50],keep)
PY
head -30 $O/bench_kernel_stats_top50.csv | cut -c1-230
