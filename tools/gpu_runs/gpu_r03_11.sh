# Round 3, call 11: host profile again with torch's intra-op threads capped at 4 (runtime.configure_inference), bench line
mkdir -p gpurun_out/r03_11
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_11
timeout 300 python tools/profile_host.py > $O/host_profile.txt 2>&1
head -40 $O/host_profile.txt | cut -c1-160
timeout 400 python bench.py --no-cpu-baseline --no-parity > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_11/bench_default.json'))
for k in ('value','ms_per_step','single_sequence_fps','fp32_exact_fps','association'):
    print(k, d.get(k))
PY
timeout 400 python bench.py --no-cpu-baseline --no-parity --no-calibration --no-roofline --no-fp32-exact > $O/bench_nocal.json 2> $O/bench_nocal.err
python tools/summarize_bench.py $O
