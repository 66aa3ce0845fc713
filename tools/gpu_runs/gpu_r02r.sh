mkdir -p gpurun_out/r02r
cd $GRAFT_REPO_ROOT
T="tests/test_models_gpu.py::test_model_forward_matches_reference_cpu_path"
(timeout 300 python -m pytest "$T" -x -q -k cfg2_deformable 2>&1 | grep -E "Mismatch|Max abs|Max rel|passed|failed" | head) > gpurun_out/r02r/a_default.log
(TF_SPLIT_LINEAR=0 timeout 300 python -m pytest "$T" -x -q -k cfg2_deformable 2>&1 | grep -E "Mismatch|Max abs|Max rel|passed|failed" | head) > gpurun_out/r02r/b_nosplit.log
(TF_NO_MHA=1 timeout 300 python -m pytest "$T" -x -q -k cfg2_deformable 2>&1 | grep -E "Mismatch|Max abs|Max rel|passed|failed" | head) > gpurun_out/r02r/c_nomha.log
(TF_LINEAR_VARIANT=0 timeout 300 python -m pytest "$T" -x -q -k cfg2_deformable 2>&1 | grep -E "Mismatch|Max abs|Max rel|passed|failed" | head) > gpurun_out/r02r/d_variant0.log
(TF_MSDA_PQUAD="on=0" timeout 300 python -m pytest "$T" -x -q -k cfg2_deformable 2>&1 | grep -E "Mismatch|Max abs|Max rel|passed|failed" | head) > gpurun_out/r02r/e_nopquad.log
(timeout 300 python -m pytest tests/test_fused_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/r02r/f_fused.log
