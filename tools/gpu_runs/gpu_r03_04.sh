# Round 3, call 4: the eight-wave variant of the encoder kernel; bench.py with calibration / fp32-exact / parity;
# a kernel trace of the cfg-3 training step.
mkdir -p gpurun_out/r03_04
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03_04
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $*" | tee -a $O/timeline.txt; }
timeout 300 tools/bin/msda_bench --iters 24 --sets 4 --patterns pert,init,local --fused 1 pquad "pquad:thr=512,npass=1,wgs=2,lds=78" "pquad:thr=512,npass=1,wgs=2,lds=64" "pquad:thr=512,npass=1,wgs=2,lds=52" "pquad:thr=512,npass=1,wgs=2,lds=78,hy=5,hx=8" > $O/pquad_threads.txt 2>&1
grep -E "fused|plain" $O/pquad_threads.txt | cut -c1-120
stamp "pquad variants"
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
cut -c1-3000 $O/bench_default.json
stamp "bench default"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof_train -- python $GRAFT_REPO_ROOT/bench.py --config cfg3 --no-cpu-baseline --no-roofline --steps 4 --warmup 2 --min-seconds 0.5 > $O/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_train -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/train_breakdown.py $f $O/train_step_breakdown.txt | head -50
rm -rf $O/prof_train
tail -2 $O/prof_train.log
stamp "done"
