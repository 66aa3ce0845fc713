# Round 6, call 20: closing counter passes -- msda_fwd_f32_pquad2 (nt stores, the default): rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE
# (own passes), SQ / LDS counters; msda_bwd_f32_sorted2 at the cfg-3 encoder shape (N = 2, local pattern): kernel stats + FETCH_SIZE / WRITE_SIZE
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_20
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/bin/msda_bench
CMD="$B --iters 24 --sets 4 --fused 1 --patterns pert pquad"
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/msda_fwd_pquad2_kernel_stats.csv
CMD="$B --iters 8 --sets 4 --fused 1 --patterns pert pquad"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
  "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/set$i -- $CMD > $O/set$i.log 2>&1
  f=$(find $O/set$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/fwd_set$i.json --match msda_fwd > /dev/null
  rm -rf $O/set$i
done
rm -rf $O/stats
CMDB="python $R/tools/bench_msda.py --no-forward --shapes cfg3_encoder_n2 --modes local --iters 10"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bstats -- $CMDB > $O/bstats.log 2>&1
f=$(find $O/bstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/msda_bwd_sorted2_kernel_stats.csv
rm -rf $O/bstats
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/bset$i -- $CMDB > $O/bset$i.log 2>&1
  f=$(find $O/bset$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f $O/bwd_set$i.json --match msda_bwd zero_words > /dev/null
  rm -rf $O/bset$i
done
cd $R
head -3 $O/msda_fwd_pquad2_kernel_stats.csv | cut -c1-200
grep -i "msda_bwd\|zero_words" $O/msda_bwd_sorted2_kernel_stats.csv | cut -c1-200
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/*set*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        print(f.split('/')[-1], k[:40], {a:(round(b/v["dispatches"]) if a!="dispatches" else b) for a,b in v.items()})
PY
