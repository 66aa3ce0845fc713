# Round 3, call 13: sequences interleaved in one thread (step_async / step_finish), 1..4 sequences per GPU; host profile
mkdir -p gpurun_out/r03_13
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_13
for seq in 1 2 3 4; do
  timeout 200 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-fp32-exact --no-single-sequence --sequences $seq > $O/bench_seq$seq.json 2> $O/bench_seq$seq.err
  tail -2 $O/bench_seq$seq.err
done
timeout 200 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-fp32-exact --no-single-sequence --no-calibration --sequences 2 > $O/bench_nocal_seq2.json 2> $O/bench_nocal_seq2.err
timeout 200 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-fp32-exact --no-single-sequence --no-calibration --sequences 4 > $O/bench_nocal_seq4.json 2> $O/bench_nocal_seq4.err
python tools/summarize_bench.py $O
timeout 300 python -m pytest tests/test_models_gpu.py -m gpu -q -x -k "tracker" 2>&1 | tail -4
