# Round 4, call 3: where the six-term dense kernels lose their time -- PMC counters (issue / issue-stalled / parked shares of the
# wave cycles, MFMA busy, LDS activity and conflicts, L1 stalls, the engine clock under matrix load) of the feed-forward kernel,
# the stream and block GEMMs and the 3 x 3 convolution at their cfg-2 shapes.
mkdir -p gpurun_out/r04_03
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/pmc_dense.sh r04_six 6 > gpurun_out/r04_03/pmc_dense_six_terms.txt 2>&1
tail -150 gpurun_out/r04_03/pmc_dense_six_terms.txt
