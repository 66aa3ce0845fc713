# Round 3, call 12: sequences per GPU x GIL switch interval with the calibrated association leg
mkdir -p gpurun_out/r03_12
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_12
for seq in 1 2 3 4 6; do
  timeout 200 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-fp32-exact --no-single-sequence --sequences $seq > $O/bench_seq$seq.json 2> $O/bench_seq$seq.err
done
for sw in 5e-3 1e-3 5e-5; do
  TF_GIL_SWITCH=$sw timeout 200 python bench.py --no-cpu-baseline --no-parity --no-roofline --no-fp32-exact --no-single-sequence --sequences 4 > $O/bench_seq4_sw$sw.json 2> $O/bench_seq4_sw$sw.err
done
python tools/summarize_bench.py $O
