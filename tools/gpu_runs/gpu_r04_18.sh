# Round 4, call 18: where a cfg-5 step goes (rocprofv3 --kernel-trace --stats of the bench command, one sequence).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_18
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --config cfg5 --no-cpu-baseline --no-parity --no-fp32-exact --no-split3 --no-single-sequence --no-roofline --sequences 1 --steps 12 --warmup 4 --min-seconds 0.5 > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 $f > $O/cfg5_kernel_stats_top45.csv
rm -rf $O/stats
tail -2 $O/stats.log | cut -c1-300
cut -c1-200 $O/cfg5_kernel_stats_top45.csv | head -36
