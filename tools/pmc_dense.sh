#!/bin/bash
# PMC counters of the dense (split-product) kernels at their cfg-2 shapes, from the standalone harnesses (no Python start-up):
#   bash tools/pmc_dense.sh <tag> [terms]       (on the GPU box, from the repo root; every --pmc set is its own run with
#                                                --kernel-trace only, as gpurun requires)
# Output: gpurun_out/pmc_<tag>/<harness>.txt = per kernel family: dispatches, average duration, every counter per dispatch,
# MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)) and the measured engine clock
# (GRBM_GUI_ACTIVE / duration).
set -u
TAG=$1
TERMS=${2:-16}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TF_SPLIT_TERMS=$TERMS
declare -A CMDS
CMDS[ffn]="$REPO/tools/bin/ffn_bench 22223 1024"
CMDS[lin256]="$REPO/tools/bin/linear_bench 22223 256 256"
CMDS[lin1024p]="$REPO/tools/bin/linear_bench 22223 1024 256 packed"
CMDS[lin256x1024p]="$REPO/tools/bin/linear_bench 22223 256 1024 packed"
CMDS[conv]="python $REPO/tools/conv3_once.py --iters 4"
for name in ${PMC_DENSE_ONLY:-ffn lin256 lin1024p lin256x1024p conv}; do
  i=0
  : > $OUT/$name.csvlist
  for SET in \
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
    "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" ; do
    i=$((i+1))
    timeout 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/${name}_set$i -- ${CMDS[$name]} > $OUT/${name}_set$i.log 2>&1
    find $OUT/${name}_set$i -name "*counter_collection.csv" >> $OUT/$name.csvlist
  done
  python3 - $OUT/$name.csvlist > $OUT/$name.txt <<'PY'
import csv, sys, collections, re
agg = collections.OrderedDict()
for path in open(sys.argv[1]).read().split():
    for r in csv.DictReader(open(path)):
        n = r.get("Kernel_Name", "")
        if not any(k in n for k in ("split_", "stream_", "ffn_fused", "linear_res_ln", "stem_conv", "conv_splitk")):
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*$", "", n).replace("void ", "")
        d = agg.setdefault(n[:90], collections.defaultdict(float))
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        d["_n_" + r["Counter_Name"]] += 1
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            d["_dur"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            d["_ndur"] += 1
for n, d in agg.items():
    us = d["_dur"] / max(d["_ndur"], 1) / 1e3
    print("%s   avg %.2f us" % (n, us))
    per = {k: v / d["_n_" + k] for k, v in d.items() if not k.startswith("_")}
    for k, v in per.items():
        print("   %-36s %16.0f per dispatch" % (k, v))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in per and us > 0:
        print("   mfma_util (2.4 GHz x 1024 SIMDs)     %16.3f" % (per["SQ_VALU_MFMA_BUSY_CYCLES"] / (us * 1e-6 * 2.4e9 * 1024)))
    if "GRBM_GUI_ACTIVE" in per and us > 0:
        print("   engine clock (GRBM_GUI_ACTIVE / t)   %13.2f GHz" % (per["GRBM_GUI_ACTIVE"] / (us * 1e-6) / 1e9))
    if "SQ_WAVE_CYCLES" in per:
        w = per["SQ_WAVE_CYCLES"]
        print("   of wave cycles: issuing %.2f, issue-stalled (WAIT_INST_ANY) %.2f, parked (WAIT_ANY) %.2f" % (
            per.get("SQ_ACTIVE_INST_ANY", 0) / w, per.get("SQ_WAIT_INST_ANY", 0) / w, per.get("SQ_WAIT_ANY", 0) / w))
PY
  rm -rf $OUT/${name}_set*/ 
done
cd $REPO
for name in ${PMC_DENSE_ONLY:-ffn lin256 lin1024p lin256x1024p conv}; do echo "#### $name"; cat $OUT/$name.txt; done
