cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r31
timeout 200 python tools/experiments/conv3_tiles.py --ti 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r31/conv3_tiles_2x2_waves.txt
cat gpurun_out/r31/conv3_tiles_2x2_waves.txt
