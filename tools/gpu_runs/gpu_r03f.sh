# PMC counters of the packed-weight GEMMs at 22 223 x 256 -> 1024 (activation-stationary TI = 3, streaming TI = 2)
mkdir -p gpurun_out/r03f
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r03f
export LD_LIBRARY_PATH=$REPO/trackformer_amd/lib:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
for v in packeda3 packed2; do
 i=0
 for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
   "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
   "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  [ $v = packed2 ] && [ $i = 3 ] && continue
  timeout 60 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/${v}_$i -- $REPO/tools/bin/linear_bench 22223 256 1024 $v > $OUT/${v}_$i.log 2>&1
  f=$(find $OUT/${v}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f $OUT/${v}_$i.json --match split_gemm_ > /dev/null
  rm -rf $OUT/${v}_$i
 done
done
