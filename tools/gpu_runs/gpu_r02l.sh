mkdir -p gpurun_out/r02l
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r02l
cd /tmp && export TMPDIR=/tmp
# steady-state per-frame breakdown: eager, one sequence
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/eager -- python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 4 --sequences 1 --no-graph --no-roofline --min-seconds 0 > $OUT/eager.log 2>&1
f=$(find $OUT/eager -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $REPO/tools/frame_breakdown.py $f $OUT/e2e_eager_per_frame.txt > /dev/null
rm -rf $OUT/eager
# matrix-core utilisation of the split GEMM at the encoder shapes
for shape in "22223 256 256" "22223 256 1024" "22223 1024 256" "400 256 256"; do
  tag=$(echo $shape | tr ' ' 'x')
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/mfma_$tag -- $REPO/tools/bin/linear_bench $shape > $OUT/mfma_$tag.log 2>&1
  f=$(find $OUT/mfma_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f $OUT/mfma_$tag.json > /dev/null
  rm -rf $OUT/mfma_$tag
done
du -sh $OUT
