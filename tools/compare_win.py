"""Compare the opt-in LDS-window encoder kernel with the default row-gather kernel (cfg-2 encoder shape)."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tools.bench_msda import make_inputs, CFG2_SHAPES
from trackformer_amd import msda, _cabi
S = sum(h*w for h, w in CFG2_SHAPES)
starts = np.cumsum([0] + [h*w for h, w in CFG2_SHAPES])
for mode in ("init", "local"):
    v, sh, loc, attn, go = make_inputs(1, 8, 32, S, 4, CFG2_SHAPES, mode, "cuda:0", encoder_refs=True)
    _cabi.lib().tf_msda_set_tiled(0)
    ref = msda.ms_deform_attn_forward(v, sh, loc, attn, 64)
    _cabi.lib().tf_msda_set_tiled(1)
    out = msda.ms_deform_attn_forward(v, sh, loc, attn, 64)
    torch.cuda.synchronize()
    d = (out - ref).abs().view(S, 8, 32).amax(-1)   # [S, M]
    bad = (d > 1e-4)
    print(mode, "max diff", d.max().item(), "bad pairs", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()
        qs = idx[:, 0].cpu().numpy()
        for l in range(4):
            sel = (qs >= starts[l]) & (qs < starts[l+1])
            print("  level", l, "bad", int(sel.sum()))
        print("  heads", torch.bincount(idx[:, 1], minlength=8).tolist())
        q0 = qs[qs < starts[1]]
        if len(q0):
            ys, xs = q0 // 167, q0 % 167
            print("  L0 y mod 12 hist", np.bincount(ys % 12, minlength=12).tolist())
            print("  L0 x mod 8 hist", np.bincount(xs % 8, minlength=8).tolist())
        print("  first", idx[:10].tolist())
