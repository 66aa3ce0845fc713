# Round 5, closing tree: matrix-core utilisation by the counters (own --pmc passes, --kernel-trace only) of the dense-kernel harnesses
# and of an eager frame -> profiles/r05_mfma_utilisation.json (what bench.py's `mfma_utilisation.pmc` reads)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r29
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/mfma_lin1 -- $R/tools/bin/linear_bench 22223 256 1024 packed > $O/mfma_lin1.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/mfma_lin2 -- $R/tools/bin/linear_bench 22223 256 256 > $O/mfma_lin2.log 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/mfma_ffn -- $R/tools/bin/ffn_bench 22223 1024 > $O/mfma_ffn.log 2>&1
for d in mfma_lin1 mfma_lin2 mfma_ffn; do
  f=$(find $O/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/$d.json > /dev/null
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/mfma_frame -- python $R/bench.py --no-cpu-baseline --no-parity --no-fp32-exact --no-split3 --no-single-sequence --no-roofline --no-graph --sequences 1 --steps 4 --warmup 2 --min-seconds 0.1 > $O/mfma_frame.log 2>&1
f=$(find $O/mfma_frame -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/mfma_frame.json > /dev/null
rm -rf $O/mfma_lin1 $O/mfma_lin2 $O/mfma_ffn $O/mfma_frame
ls -la $O
