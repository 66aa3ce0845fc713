mkdir -p gpurun_out/r02g
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# 1. kernel stats of one bench run (1 sequence so that the per-frame kernel list is clean)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r02g/bench_stats -- python $REPO/bench.py --no-cpu-baseline --sequences 1 --no-single-sequence --steps 60 > $REPO/gpurun_out/r02g/bench_prof.json 2> $REPO/gpurun_out/r02g/bench_prof.err
# 2. kernel stats + PMC of the encoder kernel through the harness (fused entry, pert, 4 sets)
CMD="$REPO/tools/bin/msda_bench --iters 8 --sets 4 --fused 1 --patterns pert quad pquad"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r02g/harness_stats -- $CMD > $REPO/gpurun_out/r02g/harness.log 2>&1
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
  "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" ; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $REPO/gpurun_out/r02g/pmc$i -- $CMD > $REPO/gpurun_out/r02g/pmc$i.log 2>&1
  f=$(find $REPO/gpurun_out/r02g/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f $REPO/gpurun_out/r02g/pmc$i.json --match msda_fwd > /dev/null
  rm -rf $REPO/gpurun_out/r02g/pmc$i
done
cd $REPO
for d in bench_stats harness_stats; do
  f=$(find gpurun_out/r02g/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r02g/${d}_kernel_stats.csv
  rm -rf gpurun_out/r02g/$d
done
du -sh gpurun_out/r02g
# 3. training bench cross-check
(timeout 300 python tools/bench_train.py 2>/dev/null) > gpurun_out/r02g/bench_train_tool.json
