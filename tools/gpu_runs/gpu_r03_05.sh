# Round 3, call 5: the hinted encoder kernel on hardware (parity + timing), bench.py with hints / calibration.
mkdir -p gpurun_out/r03_05
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03_05
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $*" | tee -a $O/timeline.txt; }
timeout 600 python -m pytest tests/test_msda_gpu.py -m gpu -q -x -k "hinted or tiled_kernel or persistent or fused_prologue" > $O/pytest_hint.txt 2>&1
tail -15 $O/pytest_hint.txt
stamp "pytest"
timeout 300 tools/bin/msda_bench --iters 24 --sets 4 --patterns pert,init,local --fused 1 pquad "pquad:hint=1" > $O/pquad_hint.txt 2>&1
grep -E "fused|plain" $O/pquad_hint.txt | cut -c1-130
stamp "msda_bench"
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_05/bench_default.json'))
for k in ('value','ms_per_step','single_sequence_fps','fp32_exact_fps','association','parity','roofline'):
    print(k, d.get(k))
PY
stamp "bench"
TF_MSDA_HINTS=0 timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-parity --no-fp32-exact > $O/bench_nohints.json 2> $O/bench_nohints.err
python tools/summarize_bench.py $O | tee $O/summary.txt
stamp "done"
