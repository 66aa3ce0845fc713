# where a cfg-5 step goes: rocprofv3 kernel statistics of the bench command with every side leg off
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r28
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $REPO/bench.py --config cfg5 --no-cpu-baseline --no-split3 --no-fp32-exact --no-parity --no-roofline --no-single-sequence --steps 24 --warmup 8 > $OUT/prof.log 2>&1
cd $REPO
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -41 $f > $OUT/cfg5_kernel_stats_top40.csv
grep -h '"metric"' $OUT/prof.log | tail -1 > $OUT/line.json
rm -rf $OUT/prof
ls -la $OUT
