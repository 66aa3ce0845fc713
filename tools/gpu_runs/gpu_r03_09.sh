# Round 3, call 9: where does the host time of a calibrated step go?
mkdir -p gpurun_out/r03_09
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_09
timeout 300 python tools/profile_host.py > $O/host_profile.txt 2>&1
head -90 $O/host_profile.txt | cut -c1-160
