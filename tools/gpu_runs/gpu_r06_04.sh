# Round 6, call 4: msda_fwd_f32_pquad2 with the conflict-free gather (default build) against the round-5 gather (variant library cf0):
# harness timing on three patterns, phase stamps of the second tile, and the LDS / VALU counters of both
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_04
mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/bin/msda_bench
timeout 200 $B --iters 24 --sets 4 --fused 1 --patterns pert,init,local --trace-dump $O/trace_cf.csv pquad pquad:ti=1 pquad:waves=8,npass=1,wgs=2,lds=78 > $O/msda_cf.txt 2>&1
grep -v "^  " $O/msda_cf.txt | cut -c1-130
echo "== cf0 (round-5 gather)"
LD_PRELOAD=$GRAFT_REPO_ROOT/tools/bin/ablate/libtf_msda_cf0.so timeout 100 $B --iters 24 --sets 4 --fused 1 --patterns pert,init,local pquad > $O/msda_cf0.txt 2>&1
grep -v "^  " $O/msda_cf0.txt | grep "pquad" | cut -c1-130
cd /tmp && export TMPDIR=/tmp
CMD="$B --iters 8 --sets 4 --fused 1 --patterns pert pquad"
for lib in cf cf0; do
  i=0
  for SET in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    if [ $lib = cf0 ]; then export LD_PRELOAD=$GRAFT_REPO_ROOT/tools/bin/ablate/libtf_msda_cf0.so; else unset LD_PRELOAD; fi
    timeout 90 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/pmc_${lib}_$i -- $CMD > $O/pmc_${lib}_$i.log 2>&1
    unset LD_PRELOAD
    f=$(find $O/pmc_${lib}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $O/pmc_${lib}_$i.json --match msda_fwd > /dev/null
    rm -rf $O/pmc_${lib}_$i
  done
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/pmc_*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        if 'pquad' in k: print(f.split('/')[-1], k[:30], {a:(round(b/v["dispatches"]) if a!="dispatches" else b) for a,b in v.items()})
PY
