#!/bin/bash
# PMC counters of split_conv3_kernel at two ResNet-50 shapes (tools/conv3_once.py); run on the GPU box from the repo root.
# Usage: bash tools/pmc_conv3.sh <tag>.  Every --pmc set is its own run with --kernel-trace only.
set -u
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/conv3_once.py --iters 4"
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
  "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16" \
  "FETCH_SIZE WRITE_SIZE TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/set$i -- $CMD > $OUT/set$i.log 2>&1
  f=$(find $OUT/set$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f $OUT/set$i.json --match split_conv3 conv_splitk > /dev/null
  [ -n "$f" ] && python3 - "$f" > $OUT/set$i.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r.get("Kernel_Name", "")
    if "split_conv3" not in n:
        continue
    key = (n[:70], r.get("Grid_Size", ""))
    d = agg.setdefault(key, collections.defaultdict(float))
    d[r["Counter_Name"]] += float(r["Counter_Value"])
    d["_n_" + r["Counter_Name"]] += 1
for (n, g), d in agg.items():
    print(n, "grid", g)
    for k, v in d.items():
        if not k.startswith("_n_"):
            print("   %-40s %16.0f per dispatch" % (k, v / d["_n_" + k]))
PY
  rm -rf $OUT/set$i
done
cd $REPO
cat $OUT/set*.txt
tail -3 $OUT/set5.log
