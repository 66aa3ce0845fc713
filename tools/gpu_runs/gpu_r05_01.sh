# Round 5, call 1 (prepared at the end of round 4, when its GPU budget was spent): what round 4 left unmeasured.
#   1. phase trace (version 2: register-held stamps) of the stream GEMM's K-slice
#   2. the full bench line of cfg 5 with the mask head on the own convolution kernels (round 4 has one short leg: 15.2 ms per step)
#   3. GPU suite + smoke on the tree
mkdir -p gpurun_out/r05_01
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_01
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
python tools/build_stream_trace.py > $O/build_trace.log 2>&1
timeout 300 python tools/stream_trace.py > $O/stream_trace.txt 2>&1
cut -c1-170 $O/stream_trace.txt
timeout 600 python bench.py --config cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_01/bench_cfg5.json'))
print('cfg5', {k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps', 'split6_fps', 'split3_fps')})
PY
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
