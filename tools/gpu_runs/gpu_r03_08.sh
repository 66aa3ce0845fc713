# Round 3, call 8: the 64-frame tracker golden, the encoder-kernel tests after the clean-up, the default bench line.
mkdir -p gpurun_out/r03_08
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03_08
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $*" | tee -a $O/timeline.txt; }
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_msda_gpu.py -m gpu -q -x -k "64_frames or tiled_kernel or persistent or fused_prologue or host" --durations=5 > $O/pytest.txt 2>&1
tail -12 $O/pytest.txt
stamp "pytest"
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_08/bench_default.json'))
for k in ('value','ms_per_step','single_sequence_fps','fp32_exact_fps','association','parity'):
    print(k, d.get(k))
print(d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
stamp "bench"
