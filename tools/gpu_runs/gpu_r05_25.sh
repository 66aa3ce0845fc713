# the per-kernel table of a frame: rocprofv3 kernel statistics of the bench command with every side leg off
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r25
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python $REPO/bench.py --no-cpu-baseline --no-split3 --no-fp32-exact --no-parity --no-roofline --steps 60 --warmup 8 > $OUT/prof_bench.log 2>&1
cd $REPO
f=$(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -61 $f > $OUT/bench_kernel_stats_top60.csv
grep -h '"metric"' $OUT/prof_bench.log | tail -1 > $OUT/bench_line_under_rocprof.json
rm -rf $OUT/prof_bench
ls -la $OUT
