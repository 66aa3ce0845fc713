# Round 5, call 4: msda_fwd_f32_pquad2 (msda_pquad2.h) on hardware: parity tests of the operator, then timing against version 1.
mkdir -p gpurun_out/r05_04
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_04
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python -m pytest tests/test_msda_gpu.py -q -x 2>&1 | tail -5 | tee $O/pytest_msda.txt
B=$R/tools/bin/msda_bench
timeout 200 $B --iters 24 --sets 4 --fused 1 --patterns pert,init,local pquad pquad:v2=0 2>&1 | tee $O/pquad_v2_vs_v1.txt | cut -c1-160
timeout 120 $B --iters 24 --sets 4 --fused 1 --trace --patterns pert pquad 2>&1 | tee $O/pquad_v2_trace.txt | cut -c1-160 | tail -22
