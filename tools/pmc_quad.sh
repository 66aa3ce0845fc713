#!/bin/bash
# PMC counters of the forward kernels at the cfg-2 encoder shape through the standalone harness
# (run on the GPU box via gpurun from the repo root).  Usage: bash tools/pmc_quad.sh <tag> <config> [pattern]
# Every --pmc set is its own run with --kernel-trace only.
set -u
TAG=$1; CFG=$2; PAT=${3:-init}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$REPO/tools/bin/msda_bench --iters 4 --fused 0 --patterns $PAT direct $CFG"
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
  "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" ; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/set$i -- $CMD > $OUT/set$i.log 2>&1
  f=$(find $OUT/set$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f $OUT/set$i.json --match msda_fwd > /dev/null
  rm -rf $OUT/set$i
done
cd $REPO
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/set*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        print(k, {a:(round(b/v["dispatches"]) if a!="dispatches" else b) for a,b in v.items()})
PY
tail -3 $OUT/set4.log
