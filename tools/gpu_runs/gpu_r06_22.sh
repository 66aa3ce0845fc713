# Round 6, call 22: the LDS-DMA GEMM (TF_LINEAR_DMA = block shape) against the stream / block forms, harness shapes of the frame
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_22
mkdir -p $O
B=tools/bin/linear_bench
for shape in "22223 256 256" "22223 256 384" "22223 256 1024" "22223 1024 256" "66800 64 256" "66800 256 64" "16700 512 128" "16700 128 512" "4200 1024 256" "4200 256 1024"; do
  echo "== $shape"
  for mode in 0 1 2 3 4; do
    TF_LINEAR_DMA=$mode timeout 60 $B $shape packed 2>&1 | grep -E "us|differ" | tr '\n' ' ' | cut -c1-200; echo " [dma=$mode]"
  done
  timeout 60 $B $shape 2>&1 | grep -E "us" | cut -c1-160
done > $O/linear_dma.txt 2>&1
cat $O/linear_dma.txt
