# Round 5, call 1 (prepared at the end of round 4, when its GPU budget was spent): what round 4 left unmeasured.
#   1. phase trace (version 2: register-held stamps) of the stream GEMM's K-slice
#   2. the full bench line of cfg 5 with the mask head on the own convolution kernels (round 4 has one short leg: 15.2 ms per step)
#   3. GPU suite + smoke on the tree
#   4. the compiled drop-in module (trackformer_amd/dropin/csrc/msda_ext.cpp) on device tensors: against the ctypes binding
mkdir -p gpurun_out/r05_01
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_01
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
python tools/build_stream_trace.py > $O/build_trace.log 2>&1
timeout 300 python tools/stream_trace.py > $O/stream_trace.txt 2>&1
cut -c1-170 $O/stream_trace.txt
timeout 600 python bench.py --config cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_01/bench_cfg5.json'))
print('cfg5', {k: d.get(k) for k in ('value', 'ms_per_step', 'single_sequence_fps', 'fp32_exact_fps', 'split6_fps', 'split3_fps')})
PY
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
python - <<'PY' 2>&1 | tail -4 | tee $O/dropin_compiled.txt
import torch
from trackformer_amd import dropin, msda
ext = dropin.install(compiled=True)
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = torch.tensor([[25, 42], [13, 21], [7, 11], [4, 6]])
S = int((shapes[:, 0] * shapes[:, 1]).sum())
N, M, D, Lq, L, P = 2, 8, 32, 300, 4, 4
for dt in (torch.float32, torch.float64):
    v = torch.randn(N, S, M, D, dtype=dt, device=dev); loc = torch.rand(N, Lq, M, L, P, 2, dtype=dt, device=dev)
    a = torch.softmax(torch.randn(N, Lq, M, L * P, dtype=dt, device=dev), -1).view(N, Lq, M, L, P)
    for shp in (shapes, shapes.to(dev)):      # host shapes / the reference's device-resident shapes
        o = ext.ms_deform_attn_forward(v, shp, loc, a, 64)
        w = msda.ms_deform_attn_forward(v, shp, loc, a, 64)
        go = torch.randn_like(o)
        g = ext.ms_deform_attn_backward(v, shp, loc, a, go, 64)
        gw = msda.ms_deform_attn_backward(v, shp, loc, a, go, 64)
        print(dt, "device shapes" if shp.is_cuda else "host shapes", "forward equal", torch.equal(o, w),
              "backward max |d|", [float((x - y).abs().max()) for x, y in zip(g, gw)])
PY
