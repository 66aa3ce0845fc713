# after the removal of the three-term bf16 mode: the whole GPU suite, smoke, the default bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r26
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/pytest_gpu_all.txt; tail -1 $OUT/smoke.txt; tail -c 600 $OUT/bench_default.json
