# Round 3, call 28: rocprofv3 kernel statistics of bench.py (graph replay, 3 interleaved sequences) after the convolution work:
# GPU-busy time per frame = sum of kernel durations / frames
mkdir -p gpurun_out/r03_28
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03_28
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 8 \
  --no-cpu-baseline --no-parity --no-roofline --no-fp32-exact --no-single-sequence > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $GRAFT_REPO_ROOT
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -46 $f > $O/bench_kernel_stats_top45.csv
[ -n "$f" ] && python3 - "$f" <<'PY' | tee $O/gpu_busy.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
pq = [r for r in rows if "msda_fwd_f32_pquad" in r["Name"]]
calls = sum(int(r["Calls"]) for r in pq)
frames = calls / 6.0            # six encoder layers per frame
print("kernel time total %.1f ms over %.0f frames (pquad calls / 6): %.3f ms GPU-busy per frame" % (tot / 1e6, frames, tot / 1e6 / frames))
for r in rows[:14]:
    print("%8.3f ms/frame  %6.1f calls/frame  %s" % (float(r["TotalDurationNs"]) / 1e6 / frames, int(r["Calls"]) / frames, r["Name"][:110]))
PY
rm -rf $O/stats
cat $O/bench_under_rocprof.json | head -c 600
