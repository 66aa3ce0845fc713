#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh            -> gpurun_out/prof_*  (copy the summaries into profiles/)
# Counter passes are separate runs with --kernel-trace only (never combined with other trace domains).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY="python $REPO/bench.py --no-cpu-baseline"
MS="python $REPO/tools/prof_msda.py --shape cfg2_encoder --iters 6"

# 1. the default bench command (HIP graph, 2 sequences): per-kernel statistics
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- $PY --steps 20 --warmup 4 > $OUT/prof_bench.log 2>&1
# 2. eager, one sequence: steady-state per-frame breakdown
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_eager -- $PY --steps 10 --warmup 4 --sequences 1 --no-graph --no-roofline > $OUT/prof_eager.log 2>&1
# 3. MSDeformAttn forward, encoder shape: counters (each --pmc set is its own run)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc_msda_sq -- $MS --mode init > $OUT/pmc_msda_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_msda_mem -- $MS --mode init > $OUT/pmc_msda_mem.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_msda_fetch -- $MS --mode init > $OUT/pmc_msda_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_msda_write -- $MS --mode init > $OUT/pmc_msda_write.log 2>&1
# 4. matrix-core utilisation of the dense kernels (GEMMs, convolutions) in the eager frame
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_mfma -- $PY --steps 4 --warmup 2 --sequences 1 --no-graph --no-roofline > $OUT/pmc_mfma.log 2>&1

cd $REPO
for d in pmc_msda_sq pmc_msda_mem pmc_msda_fetch pmc_msda_write; do
    f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py $f $OUT/$d.json --match msda_fwd > /dev/null
done
f=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py $f $OUT/pmc_mfma.json > /dev/null
f=$(find $OUT/prof_eager -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/frame_breakdown.py $f $OUT/e2e_eager_per_frame.txt > /dev/null
f=$(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 $f > $OUT/bench_kernel_stats_top40.csv
# raw traces are large: keep the summaries only
rm -rf $OUT/prof_eager $OUT/pmc_mfma/*/*kernel_trace.csv 2>/dev/null
ls -la $OUT | head -40
