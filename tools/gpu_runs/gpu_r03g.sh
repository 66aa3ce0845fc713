# activation-stationary K = 256 kernel: parity (bit pattern) + timing
mkdir -p gpurun_out/r03g
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/trackformer_amd/lib:$LD_LIBRARY_PATH
{
for shape in "22223 256 256" "22223 256 384" "22223 256 1024"; do
  for v in packeda2 packeda3; do timeout 60 tools/bin/linear_bench $shape $v; done
done
} > gpurun_out/r03g/linear_astat.txt 2>&1
