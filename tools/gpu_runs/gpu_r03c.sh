# packed-weight split GEMM: parity (bit pattern vs the unpacked kernel) + timing at the encoder shapes, then its pytest cases
mkdir -p gpurun_out/r03c
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/trackformer_amd/lib:$LD_LIBRARY_PATH
{
for shape in "22223 256 256" "22223 256 384" "22223 256 1024" "22223 1024 256"; do
  timeout 60 tools/bin/linear_bench $shape
  for v in packed2 packed3 packed4; do timeout 60 tools/bin/linear_bench $shape $v; done
done
} > gpurun_out/r03c/linear_packed.txt 2>&1
timeout 200 python3 -m pytest tests/test_linear_split_gpu.py -m gpu -x -q -k "packed" > gpurun_out/r03c/pytest_packed.txt 2>&1
tail -3 gpurun_out/r03c/pytest_packed.txt
