#!/bin/bash
# Evidence run for the forward kernels at the cfg-2 encoder shape through the standalone harness (gpurun, from
# the repo root): parity + timing table, rocprofv3 kernel statistics, HBM traffic counters (one --pmc set per run).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r01q
mkdir -p $OUT
B=$REPO/tools/bin/msda_bench
timeout 15 $B --iters 20 --fused 1 --patterns init,local,uniform direct win quad > $OUT/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 40 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --iters 20 --fused 0 --patterns init direct quad > $OUT/stats.log 2>&1
timeout 40 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --iters 4 --fused 0 --patterns init direct quad > $OUT/fetch.log 2>&1
timeout 40 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B --iters 4 --fused 0 --patterns init direct quad > $OUT/write.log 2>&1
cd $REPO
for d in fetch write; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f $OUT/$d.json --match msda_fwd > /dev/null
done
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/fetch $OUT/write
cat $OUT/bench.log; cat $OUT/kernel_stats.csv; cat $OUT/fetch.json $OUT/write.json
