# Round 4, call 11: the weight-fragment ring of the stream GEMM's convolution form (4 stages against 2): per-layer table + bench.
mkdir -p gpurun_out/r04_11
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_11
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
{
for W in 2 4; do
  echo "## TF_LINEAR_STREAM_WSTAGES=$W"
  TF_LINEAR_STREAM_WSTAGES=$W timeout 200 python tools/bench_conv.py 2>&1 | grep -E "conv|downsample|per frame"
done
} > $O/conv_per_layer.txt 2>&1
grep -E "##|per frame|conv2" $O/conv_per_layer.txt | cut -c1-110
for W in 2 4; do
  TF_LINEAR_STREAM_WSTAGES=$W timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-parity --no-fp32-exact --no-split3 2> /dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('wstages $W', d['value'], d['ms_per_step'], d['single_sequence_fps'])"
done | tee $O/bench_wstages.txt
timeout 300 python -m pytest tests/test_full_size_gpu.py -m gpu -q -x -k "conv3x3_split_at_resnet_shapes or conv_split" 2>&1 | tail -2 | tee $O/pytest_conv.txt
