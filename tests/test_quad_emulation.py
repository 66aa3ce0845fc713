"""CPU: the data-dependent logic of msda_fwd_f32_quad (tile partition, extended-coordinate windows, staging
rounds, all-or-nothing capacity rule, per-point fallback) emulated on the host with the kernel's own geometry
functions (trackformer_amd/csrc/msda_quad_geom.h via tests/emu/quad_emu.cpp) and compared with the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import msda_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
CFG2 = [(100, 167), (50, 84), (25, 42), (13, 21)]


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "emu", "quad_emu.cpp")
    out_dir = os.path.join(HERE, "emu", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libquad_emu.so")
    hdr = os.path.join(os.path.dirname(HERE), "trackformer_amd", "csrc", "msda_quad_geom.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.quad_emu_forward.restype = ctypes.c_int
    lib.quad_emu_forward.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 12 + [ctypes.c_void_p]
    return lib


def make_inputs(shapes, mode, N=1, M=8, seed=0):
    """Encoder-shaped inputs: one query per pyramid pixel; sampling patterns as in tools/bench_msda.py."""
    rng = np.random.default_rng(seed)
    L, P, D = len(shapes), 4, 32
    S = sum(h * w for h, w in shapes)
    value = rng.standard_normal((N, S, M, D), dtype=np.float32)
    ref = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2)
                          for h, w in shapes]).astype(np.float32)
    ref = np.broadcast_to(ref[None, :, None, None, None, :], (N, S, M, L, P, 2))
    hw = np.array(shapes, np.float32)[None, None, None, :, None, :]          # (H, W) per level
    if mode == "init":      # 8-direction bias grid of MSDeformAttn._reset_parameters, (H, W) divisor as written
        dirs = np.array([(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1) if (a, b) != (0, 0)], np.float32)
        k = np.arange(1, P + 1, dtype=np.float32)[None, None, None, None, :, None]
        loc = ref + dirs[:M][None, None, :, None, None, :] * k / hw
    elif mode == "local":   # reference point + N(0, 2 px)
        loc = ref + rng.standard_normal((N, S, M, L, P, 2), dtype=np.float32) * 2.0 / hw[..., ::-1]
    elif mode == "border":  # a wide spread: many samples outside the level, many points leave their window
        loc = ref + rng.standard_normal((N, S, M, L, P, 2), dtype=np.float32) * 0.2
    else:
        loc = rng.random((N, S, M, L, P, 2), dtype=np.float32)
    attn = rng.random((N, S, M, L * P), dtype=np.float32)
    attn = (attn / attn.sum(-1, keepdims=True)).reshape(N, S, M, L, P)
    return value, np.ascontiguousarray(loc, np.float32), np.ascontiguousarray(attn, np.float32)


def run(emu, shapes, value, loc, attn, th, tw, hy=6, hx=10, cap_rows=312, cap_q=192, round0=1, ta=0):
    N, S, M, D = value.shape
    shp = np.array(shapes, np.int64)
    out = np.full((N, S, M * D), np.nan, np.float32)
    stats = np.zeros(5, np.int64)
    rc = emu.quad_emu_forward(value.ctypes.data, shp.ctypes.data, loc.ctypes.data, attn.ctypes.data, out.ctypes.data,
                              N, S, M, len(shapes), th, tw, hy, hx, cap_rows, cap_q, round0, ta, stats.ctypes.data)
    assert rc == 0, rc
    ref = msda_oracle.msda_forward(value, shp, loc, attn, nthreads=8)
    np.testing.assert_allclose(out, ref.reshape(out.shape), atol=1e-5, rtol=1e-4)
    return stats


CASES = [
    # name, shapes, mode, N, tile, kwargs
    ("cfg2_init_default_plan", CFG2, "init", 1, (10, 14), {}),
    ("cfg2_local_default_plan", CFG2, "local", 1, (10, 14), {}),
    ("cfg2_uniform_no_locality", CFG2, "uniform", 1, (10, 14), {}),
    ("cfg2_init_one_round", CFG2, "init", 1, (8, 12), dict(round0=0xF, cap_rows=504, cap_q=128)),
    ("cfg2_init_ta12", CFG2, "init", 1, (8, 12), dict(round0=0xF, ta=12, cap_rows=312, cap_q=128)),
    ("cfg2_init_ta8_two_rounds", CFG2, "init", 1, (8, 12), dict(round0=1, ta=8, cap_rows=248, cap_q=128)),
    ("small_pyramid_n2_border", [(25, 42), (13, 21), (7, 11), (4, 6)], "border", 2, (8, 12), {}),
    ("small_pyramid_tight_halo", [(25, 42), (13, 21), (7, 11), (4, 6)], "local", 1, (8, 12), dict(hy=1, hx=1)),
    ("small_pyramid_tiny_capacity", [(25, 42), (13, 21), (7, 11), (4, 6)], "init", 1, (8, 12), dict(cap_rows=40)),
    ("tiny_levels", [(3, 5), (2, 3), (1, 2), (1, 1)], "local", 1, (3, 5), {}),
    ("one_level", [(37, 53)], "local", 1, (8, 16), {}),
    ("two_levels_odd_tiles", [(40, 60), (20, 30)], "init", 1, (7, 9), {}),
]


@pytest.mark.parametrize("name,shapes,mode,N,tile,kw", CASES, ids=[c[0] for c in CASES])
def test_emulated_kernel_matches_oracle(emu, name, shapes, mode, N, tile, kw):
    value, loc, attn = make_inputs(shapes, mode, N=N, seed=len(name))
    stats = run(emu, shapes, value, loc, attn, tile[0], tile[1], **kw)
    staged, left_window, by_loads, tiles, max_q = (int(x) for x in stats)
    if name == "cfg2_init_default_plan":
        # the bench pattern: every window fits, (almost) every point is served from LDS
        assert by_loads == 0 and left_window == 0 and staged > 0 and max_q <= 192
    if name == "cfg2_uniform_no_locality":
        # no locality: the fine levels' windows overflow (whole level by buffer loads); the coarse levels fit
        # their clamped windows, and most of their points lie outside them (per-point fallback)
        assert by_loads > 0 and left_window > staged
    if name == "small_pyramid_tight_halo":
        assert left_window > 0      # points outside the clamped window take the per-point fallback
    if name == "small_pyramid_tiny_capacity":
        assert by_loads > 0


def test_window_geometry_is_overflow_safe():
    """tfq_window with wild bounding boxes (what a broken reduction would deliver) never yields a window larger
    than the nominal footprint -- the property the kernel's staging loop bound relies on."""
    import itertools
    src = r'''
    #include "../../trackformer_amd/csrc/msda_quad_geom.h"
    extern "C" int probe(int bx0, int bx1, int by0, int by1, int* o) {
        bool fits; QuadWindow q = tfq_window(bx0, bx1, by0, by1, -1, 30, -1, 20, 1000, 2, &fits);
        o[0] = q.ww; o[1] = q.wh; o[2] = fits; return q.ww * q.wh; }
    '''
    d = os.path.join(HERE, "emu", "_build")
    os.makedirs(d, exist_ok=True)
    c = os.path.join(HERE, "emu", "_probe.cpp")
    open(c, "w").write(src)
    so = os.path.join(d, "libprobe.so")
    try:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", c, "-o", so])
    finally:
        os.remove(c)
    lib = ctypes.CDLL(so)
    o = (ctypes.c_int * 3)()
    wild = [-2**31, -2**31 + 1, -5, -1, 0, 7, 29, 30, 2**31 - 2, 2**31 - 1]
    for bx0, bx1, by0, by1 in itertools.product(wild, repeat=4):
        n = lib.probe(bx0, bx1, by0, by1, o)
        assert 0 <= n <= 32 * 22 and 0 <= o[0] <= 32 and 0 <= o[1] <= 22
