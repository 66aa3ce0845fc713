# Round 3, call 23: PMC counters of the 3x3 split-product convolution (layer1 and layer3 shapes), split-K policy 768,300,8,32
cd $GRAFT_REPO_ROOT
TF_CONV_KSPLIT_POLICY=768,300,8,32 bash tools/pmc_conv3.sh r03_conv3
