# Round 5, call 2: the round's new parity tests on hardware (well-conditioned 64-frame sequences, cfg-4 tracker, compiled drop-in on
# device tensors, fp16 range tools, rank verification), the whole GPU suite, the pquad baseline of the day, the bench line with the
# one-sequence headline.
mkdir -p gpurun_out/r05_02
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_02
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -30 | tee $O/pytest_gpu_all.txt
B=$R/tools/bin/msda_bench
timeout 120 $B --iters 24 --sets 4 --fused 1 --patterns pert,init,local pquad 2>&1 | tee $O/pquad_baseline.txt | cut -c1-160
timeout 120 $B --iters 24 --sets 4 --fused 1 --trace --patterns pert pquad 2>&1 | tee $O/pquad_trace.txt | cut -c1-160 | tail -20
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_02/bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'multi_sequence_fps', 'fp32_exact_fps', 'single_sequence_fp32_exact_fps', 'split6_fps', 'split3_fps')})
print(d['parity']); print(d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('traffic'))
PY
