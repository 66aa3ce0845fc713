# closing evidence of round 5: the whole GPU suite, smoke, the bench lines, rocprofv3 kernel statistics of the bench command and of
# the roofline measurement alone
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r24
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in cfg3 cfg4 cfg5 cfg1; do timeout 900 python bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python $REPO/bench.py --no-cpu-baseline --no-split3 --steps 20 --warmup 4 > $OUT/prof_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_roofline -- python $REPO/bench.py --roofline-only > $OUT/prof_roofline.log 2>&1
cd $REPO
f=$(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -61 $f > $OUT/bench_kernel_stats_top60.csv
f=$(find $OUT/prof_roofline -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 $f > $OUT/bench_roofline_kernel_stats.csv
rm -rf $OUT/prof_bench $OUT/prof_roofline
tail -3 $OUT/pytest_gpu_all.txt; tail -1 $OUT/smoke.txt; ls -la $OUT
