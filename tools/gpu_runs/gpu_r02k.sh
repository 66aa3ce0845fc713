mkdir -p gpurun_out/r02k
cd $GRAFT_REPO_ROOT
(timeout 300 python tools/profile_host.py 2>&1 | tail -60) > gpurun_out/r02k/host_profile.txt
(TF_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-roofline --no-single-sequence 2>gpurun_out/r02k/multi.err) > gpurun_out/r02k/multi.json
(timeout 900 python -m pytest tests/test_msda_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/r02k/pytest_msda.log
