# Round 4, call 13: stream GEMM with the split of the next slice interleaved with this slice's MFMAs (TF_LINEAR_STREAM_INTERLEAVE=1) against the default.
mkdir -p gpurun_out/r04_13
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_13
cd $R
{
for W in 0 1; do
  echo "## TF_LINEAR_STREAM_INTERLEAVE=$W"
  TF_LINEAR_STREAM_INTERLEAVE=$W timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv|downsample|per frame"
  for s in "22223 256 1024 packed" "22223 1024 256 packed" "22223 256 256 packed" "16700 512 128 packed"; do
    echo "# $s"; TF_LINEAR_STREAM_INTERLEAVE=$W timeout 60 tools/bin/linear_bench $s 2>&1 | grep -E "us per launch|differ"
  done
done
} > $O/interleave.txt 2>&1
grep -E "##|per frame|conv2|^# |us per launch|differ" $O/interleave.txt | sed 's/ = .*TFLOP.*//' | cut -c1-110
