mkdir -p gpurun_out/r02w
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r02w
cd $REPO
(timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -5) > $OUT/pytest_gpu_all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roof -- python $REPO/bench.py --roofline-only > $OUT/roofline_only.json 2> $OUT/roofline_only.err
f=$(find $OUT/roof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/bench_roofline_kernel_stats.csv
rm -rf $OUT/roof
cd $REPO
(timeout 600 python bench.py --no-cpu-baseline 2>/dev/null) > $OUT/bench_default_nocpu.json
