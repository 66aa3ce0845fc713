# Round 4, call 12: counters of the dense kernels with the fp16 split product (convolutions through the stream GEMM, the
# feed-forward block, the 256 -> 1024 linear): what a launch waits for now that a product is three MFMAs.
R=$GRAFT_REPO_ROOT
cd $R
PMC_DENSE_ONLY="conv ffn lin256x1024p lin256" bash tools/pmc_dense.sh r04_12 16
for n in conv ffn lin256x1024p lin256; do echo "#### $n"; cat gpurun_out/pmc_r04_12/$n.txt; done > gpurun_out/pmc_r04_12/all.txt
rm -rf gpurun_out/pmc_r04_12/*_set*
grep -E "####|avg|mfma_util|of wave cycles" gpurun_out/pmc_r04_12/all.txt | cut -c1-160
