# Round 6, call 17: ordered kernel sequence of the image-only half of a cfg-2 frame (which launches are not the library's own?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_17
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/tools/experiments/encode_sequence.py > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/experiments/encode_sequence_report.py $f > $O/encode_sequence.txt
rm -rf $O/prof
grep -c . $O/encode_sequence.txt; grep -v "stream_gemm\|split_gemm\|ffn_fused\|msda_fwd\|linear_res_ln\|halo\|splitk_reduce" $O/encode_sequence.txt | cut -c1-140
