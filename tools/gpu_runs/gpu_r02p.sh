mkdir -p gpurun_out/r02p
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r02p
cd /tmp && export TMPDIR=/tmp
export TF_LINEAR_WS_SPLIT=1
for v in 2 6; do
 i=0
 for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
   "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
   "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
   "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC" ; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/v${v}_$i -- $REPO/tools/bin/linear_bench 22223 256 256 $v > $OUT/v${v}_$i.log 2>&1
  f=$(find $OUT/v${v}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f $OUT/v${v}_$i.json --match split_gemm > /dev/null
  rm -rf $OUT/v${v}_$i
 done
done
