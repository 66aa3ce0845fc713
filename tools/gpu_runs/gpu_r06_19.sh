# Round 6, call 19: where does bench.py crash with the parallel branches on?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_19
mkdir -p $O
PYTHONFAULTHANDLER=1 AMD_LOG_LEVEL=0 timeout 600 python bench.py --no-cpu-baseline --no-fp32-exact --no-split3 --no-roofline --sequences 1 --no-parity > $O/bench.json 2> $O/bench.err
tail -60 $O/bench.err | cut -c1-200
