"""CPU: the repo's HIP kernel SOURCES (trackformer_amd/csrc/*.hip) compiled for the host and executed by the SIMT emulator
of tests/emu/hipemu/ (fibers per work-item, wave operations executed in lock-step, LDS arena, buffer bounds checking,
asynchronous LDS-DMA, MFMA fragment layouts), called through the same C ABI as libtf_msda.so and compared with the
oracle / float64 references.  This is what lets kernel changes be checked without a GPU; the `-m gpu` tests remain the
parity tests proper (same entry points on the real hardware).

Every test also requires that no cross-lane read hit an inactive lane (inactive_reads); the LDS-window kernels, whose
DPP / readlane exchanges are meant to run with full waves, additionally that no wave operation was reached by only a
part of its wave (divergent_ops)."""
import numpy as np
import pytest

from oracle import msda_oracle
from tests import emu_lib
from tests.test_quad_emulation import make_inputs as encoder_inputs
from tests.util_msda import discontinuity_mask, golden_cases, load_case, rand_inputs

pytestmark = pytest.mark.skipif(not emu_lib.available(), reason="needs a host clang++ (ROCm's llvm) to build the emulated library")


@pytest.fixture(autouse=True)
def _convergent():
    emu_lib.lib()
    emu_lib.stats(reset=True)
    yield
    st = emu_lib.stats()
    assert st["inactive_reads"] == 0, st


def _np(ts):
    return [t.numpy() for t in ts]


SMALL_GOLDEN = [p for p in golden_cases() if load_case(p)["value"].size <= 200_000]


@pytest.mark.parametrize("path", SMALL_GOLDEN, ids=lambda p: p.split("msda_")[-1][:-4])
def test_golden_vectors_forward_and_backward(path):
    """The reference's own outputs (tests/golden/msda_*.npz) through the emulated kernels."""
    z = load_case(path)
    f64 = z["value"].dtype == np.float64
    atol, rtol = (1e-12, 1e-10) if f64 else (1e-5, 1e-4)
    out = emu_lib.msda_forward(z["value"], z["shapes"], z["loc"], z["attn"])
    np.testing.assert_allclose(out, z["out"], atol=atol, rtol=rtol)
    gv, gl, ga = emu_lib.msda_backward(z["value"], z["shapes"], z["loc"], z["attn"], z["grad_out"])
    np.testing.assert_allclose(gv, z["grad_value"], atol=atol * 2, rtol=rtol)
    np.testing.assert_allclose(ga, z["grad_attn"], atol=atol * 10, rtol=rtol)
    keep = ~discontinuity_mask(z["loc"], z["shapes"])
    np.testing.assert_allclose(gl[keep], z["grad_loc"][keep], atol=atol * 10, rtol=rtol)
    assert np.all(gl[~keep] == 0)


ORACLE_CASES = [
    ("tiny", dict(N=2, M=2, D=4, Lq=3, P=2, shapes=[(8, 8), (4, 4), (2, 2)])),
    ("ragged_levels", dict(N=1, M=3, D=8, Lq=65, P=3, shapes=[(1, 1), (1, 7), (9, 1), (3, 5)], loc_mode="wide")),
    ("d5_scalar_path", dict(N=2, M=3, D=5, Lq=33, P=2, shapes=[(7, 3), (2, 2)], loc_mode="wide")),
    ("d36_l8_buf", dict(N=1, M=8, D=36, Lq=70, P=4, shapes=[(13, 21), (7, 11), (4, 6), (2, 3)] * 2, loc_mode="wide")),
    ("d32_l4_direct", dict(N=2, M=8, D=32, Lq=150, P=4, shapes=[(25, 42), (13, 21), (7, 11), (4, 6)], loc_mode="local")),
    ("d64", dict(N=1, M=4, D=64, Lq=50, P=4, shapes=[(12, 10), (6, 5)], loc_mode="wide")),
    ("levels16", dict(N=1, M=2, D=8, Lq=20, P=1, shapes=[(3, 2)] * 16, loc_mode="wide")),
]


@pytest.mark.parametrize("name,kw", ORACLE_CASES, ids=[c[0] for c in ORACLE_CASES])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_forward_backward_vs_oracle(name, kw, dtype):
    """Decoder-shaped calls: msda_fwd_f32_direct / _buf / the row-gather kernels and the atomic backward kernels."""
    import torch
    value, shapes, loc, attn, grad_out = _np(rand_inputs(seed=100 + len(name), dtype=torch.float32 if dtype == "f32" else torch.float64, **kw))
    atol, rtol = (1e-5, 1e-4) if dtype == "f32" else (1e-12, 1e-10)
    ref_out = msda_oracle.msda_forward(value, shapes, loc, attn)
    rv, rl, ra = msda_oracle.msda_backward(value, shapes, loc, attn, grad_out)
    np.testing.assert_allclose(emu_lib.msda_forward(value, shapes, loc, attn), ref_out, atol=atol, rtol=rtol)
    np.testing.assert_allclose(emu_lib.msda_forward(value, shapes, loc, attn, dshapes=True), ref_out, atol=atol, rtol=rtol)
    gv, gl, ga = emu_lib.msda_backward(value, shapes, loc, attn, grad_out)
    np.testing.assert_allclose(gv, rv, atol=atol * 4, rtol=rtol)
    np.testing.assert_allclose(gl, rl, atol=atol * 20, rtol=rtol)
    np.testing.assert_allclose(ga, ra, atol=atol * 10, rtol=rtol)


# ---- encoder-shaped calls: the LDS-window kernels ------------------------------------------------------------------
PYR = [(25, 42), (13, 21), (7, 11), (4, 6)]
ENC_CASES = [
    ("pyramid_init", PYR, "init", 1, 32),
    ("pyramid_local_n2", PYR, "local", 2, 32),
    ("pyramid_uniform_fallbacks", PYR, "uniform", 1, 32),
    ("pyramid_border", PYR, "border", 1, 32),
    ("tiny_levels", [(3, 5), (2, 3), (1, 2), (1, 1)], "local", 1, 32),
    ("one_level", [(19, 23)], "local", 1, 32),
    ("coarse_first", [(7, 11), (25, 42)], "local", 1, 32),
]


def _with_d(value, D):
    if value.shape[-1] == D:
        return value
    rng = np.random.default_rng(value.shape[1])
    return rng.standard_normal(value.shape[:3] + (D,), dtype=np.float32)


@pytest.fixture(params=["quad", "pquad", "pquad_v1"])
def tiled(request):
    """quad: one tile per workgroup; pquad: the persistent kernel (version 2 where it applies: msda_pquad2.h); pquad_v1: the
    first version everywhere."""
    L = emu_lib.lib()
    prev_t = L.tf_msda_set_tiled(2)
    prev_p = L.tf_msda_set_option(b"pquad", 0 if request.param == "quad" else 1)
    prev_v = L.tf_msda_set_option(b"pquad_v2", 0 if request.param == "pquad_v1" else 1)
    yield request.param
    L.tf_msda_set_tiled(prev_t)
    L.tf_msda_set_option(b"pquad", prev_p)
    L.tf_msda_set_option(b"pquad_v2", prev_v)


@pytest.mark.parametrize("name,shapes,mode,N,D", ENC_CASES, ids=[c[0] for c in ENC_CASES])
def test_encoder_window_kernels_vs_oracle(tiled, name, shapes, mode, N, D):
    value, loc, attn = encoder_inputs(shapes, mode, N=N, seed=len(name))
    shp = np.array(shapes, np.int64)
    out = emu_lib.msda_forward(value, shp, loc, attn)
    ref = msda_oracle.msda_forward(value, shp, loc, attn, nthreads=4)
    np.testing.assert_allclose(out, ref, atol=1e-5, rtol=1e-4)
    st = emu_lib.stats()
    assert st["divergent_ops"] == 0, st
    if mode in ("init", "local") and len(shapes) > 1 and shapes[0][0] > shapes[1][0]:
        assert st["lds_dma_bytes"] > 0   # the windows were staged by LDS-DMA: it really was the window kernel


def _fused_case(shapes, N, D, seed, M=8, P=4, spread=2.0):
    rng = np.random.default_rng(seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = rng.standard_normal((N, S, M, D), dtype=np.float32)
    qproj = rng.standard_normal((N, S, 3 * M * L * P), dtype=np.float32)
    qproj[..., :2 * M * L * P] *= spread
    refp = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2)
                           for h, w in shapes]).astype(np.float32)
    refp = np.ascontiguousarray(np.broadcast_to(refp[None, :, None, :], (N, S, L, 2)))
    off = qproj[..., :2 * M * L * P].reshape(N, S, M, L, P, 2)
    logits = qproj[..., 2 * M * L * P:].reshape(N, S, M, L * P)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    attn = (e / e.sum(-1, keepdims=True)).reshape(N, S, M, L, P).astype(np.float32)
    hw = np.array(shapes, np.float32)[None, None, None, :, None, :]
    loc = (refp[:, :, None, :, None, :] + off / hw).astype(np.float32)
    return value, refp, qproj, loc, attn


PQUAD_VARIANTS = [dict(), dict(pquad_npass=1, pquad_wg_per_cu=4), dict(pquad_npass=3, pquad_wg_per_cu=2),
                  dict(pquad_wide=0), dict(pquad_lds_kb=24),
                  dict(pquad_tile_h=4, pquad_tile_w=16, pquad_wg_per_cu=1),
                  # version 2 (the default where it applies) under other plans, and version 1 at the default plan
                  dict(pquad_v2=0), dict(pquad_v2=0, pquad_lds_kb=24), dict(pquad_wg_per_cu=1), dict(pquad_halo_y=2, pquad_halo_x=2),
                  dict(pquad_lds_kb=12),
                  # eight-wave workgroups of version 2: one pass of 128 pairs, two workgroups per CU
                  dict(pquad_waves=8, pquad_npass=1, pquad_wg_per_cu=2, pquad_lds_kb=78), dict(pquad_waves=8, pquad_npass=1, pquad_wg_per_cu=1, pquad_lds_kb=30),
                  # round 6: the conflict-free gather (lanes of a quad split by tap column), small windows (the buffer-load path of the
                  # same lane mapping), eight waves; the other output-store policies and the rotated head map
                  dict(pquad_cf=1), dict(pquad_cf=1, pquad_lds_kb=12), dict(pquad_cf=1, pquad_waves=8, pquad_npass=1, pquad_wg_per_cu=2, pquad_lds_kb=78),
                  dict(pquad_store=0, pquad_headmix=1, pquad_ldnt=1), dict(pquad_store=2, pquad_headmix=2, pquad_prio=2)]


@pytest.mark.parametrize("opts", PQUAD_VARIANTS, ids=["-".join("%s%d" % (k[6:], v) for k, v in o.items()) or "default"
                                                      for o in PQUAD_VARIANTS])
def test_persistent_encoder_kernel_variants(opts):
    """msda_fwd_f32_pquad with its knobs away from the defaults, plain and fused entry, N = 2 (the GPU test of the same
    name at a smaller size): the tile loop, the double-buffered per-tile tables and the prefetch must not change a result."""
    prev = emu_lib.set_options(**opts)
    try:
        shapes = [(20, 31), (10, 16), (5, 8), (3, 4)]
        shp = np.array(shapes, np.int64)
        value, loc, attn = encoder_inputs(shapes, "local", N=2, seed=11)
        np.testing.assert_allclose(emu_lib.msda_forward(value, shp, loc, attn),
                                   msda_oracle.msda_forward(value, shp, loc, attn, nthreads=4), atol=1e-5, rtol=1e-4)
        value, refp, qproj, floc, fattn = _fused_case(shapes, 2, 32, seed=5)
        got = emu_lib.msda_forward_fused(value, shp, refp, qproj, 8, len(shapes), 4)
        np.testing.assert_allclose(got, msda_oracle.msda_forward(value, shp, floc, fattn, nthreads=4), atol=2e-5, rtol=1e-4)
        assert emu_lib.stats()["lds_dma_bytes"] > 0 and emu_lib.stats()["divergent_ops"] == 0
    finally:
        emu_lib.set_options(**prev)


@pytest.mark.parametrize("shapes,N", [(PYR, 1), ([(15, 22), (8, 11), (4, 6)], 2), ([(19, 23)], 1)],
                         ids=["four_levels", "three_levels_n2", "one_level"])
def test_persistent_encoder_kernel_head_dim_36(shapes, N):
    """hidden 288 (cfg 4): 144-byte rows packed in LDS, 3 lanes x 12 channels per pair; plain and fused entry."""
    shp = np.array(shapes, np.int64)
    value, loc, attn = encoder_inputs(shapes, "local", N=N, seed=7)
    value = _with_d(value, 36)
    np.testing.assert_allclose(emu_lib.msda_forward(value, shp, loc, attn),
                               msda_oracle.msda_forward(value, shp, loc, attn, nthreads=4), atol=1e-5, rtol=1e-4)
    assert emu_lib.stats()["lds_dma_bytes"] > 0
    value, refp, qproj, floc, fattn = _fused_case(shapes, N, 36, seed=9)
    got = emu_lib.msda_forward_fused(value, shp, refp, qproj, 8, len(shapes), 4)
    np.testing.assert_allclose(got, msda_oracle.msda_forward(value, shp, floc, fattn, nthreads=4), atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize("name,shapes,mode,N,D", [c for c in ENC_CASES if c[0] in ("pyramid_init", "pyramid_uniform_fallbacks",
                                                                                    "tiny_levels", "one_level", "coarse_first")],
                         ids=lambda v: v if isinstance(v, str) and "_" in v else None)
def test_encoder_shape_backward_sorted_kernel(name, shapes, mode, N, D):
    """msda_bwd_f32_sorted (counting sort of the taps by destination row in LDS, one full-row atomic per row)."""
    value, loc, attn = encoder_inputs(shapes, mode, N=N, seed=len(name))
    shp = np.array(shapes, np.int64)
    rng = np.random.default_rng(3)
    grad_out = rng.standard_normal((N, value.shape[1], value.shape[2] * value.shape[3]), dtype=np.float32)
    if name == "pyramid_init":   # mix in points far outside their windows
        loc = loc.copy()
        loc[:, :, :, :, ::2, 0] += 12.0 / 42
        loc[:, :, :, :, ::2, 1] -= 9.0 / 25
    gv, gl, ga = emu_lib.msda_backward(value, shp, loc, attn, grad_out)
    rv, rl, ra = msda_oracle.msda_backward(value, shp, loc, attn, grad_out)
    np.testing.assert_allclose(gv, rv, atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(gl, rl, atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(ga, ra, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("D,shapes", [(32, PYR), (36, [(13, 21), (7, 11), (4, 6), (2, 3)] * 2)], ids=["d32_direct", "d36_l8_buf"])
def test_fused_prologue_decoder_shapes(ref_dim, D, shapes):
    """tf_msda_forward_fused_f32 at decoder shapes (softmax + location arithmetic inside msda_fwd_f32_direct / _buf)."""
    rng = np.random.default_rng(7)
    N, M, L, P, Lq = 1, 8, len(shapes), 4, 77
    S = sum(h * w for h, w in shapes)
    value = rng.standard_normal((N, S, M, D), dtype=np.float32)
    qproj = rng.standard_normal((N, Lq, 3 * M * L * P), dtype=np.float32)
    ref = (rng.random((N, Lq, L, ref_dim), dtype=np.float32) * 0.6 + 0.1).astype(np.float32)
    off = qproj[..., :2 * M * L * P].reshape(N, Lq, M, L, P, 2)
    logits = qproj[..., 2 * M * L * P:].reshape(N, Lq, M, L * P)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    attn = (e / e.sum(-1, keepdims=True)).reshape(N, Lq, M, L, P).astype(np.float32)
    if ref_dim == 2:
        loc = ref[:, :, None, :, None, :] + off / np.array(shapes, np.float32)[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    shp = np.array(shapes, np.int64)
    expect = msda_oracle.msda_forward(value, shp, loc.astype(np.float32), attn)
    got = emu_lib.msda_forward_fused(value, shp, ref, qproj, M, L, P)
    np.testing.assert_allclose(got, expect, atol=2e-5, rtol=1e-4)


# ---- fused element-wise kernels, linears on the matrix cores, query self-attention ------------------------------------
def test_bias_act_and_add_layernorm():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((37, 5, 64), dtype=np.float32)
    b = rng.standard_normal(64, dtype=np.float32)
    r = rng.standard_normal(x.shape, dtype=np.float32)
    np.testing.assert_allclose(emu_lib.bias_act(x, b, r, relu=True), np.maximum(x + b + r, 0), atol=1e-6)
    np.testing.assert_allclose(emu_lib.bias_act(x, b, None, relu=False), x + b, atol=1e-6)
    big = rng.standard_normal((3000, 7, 64), dtype=np.float32)   # more float4s than the grid has threads: the strided loop
    assert np.array_equal(emu_lib.bias_act(big, b, None, relu=True), np.maximum(big + b, 0))
    assert np.array_equal(emu_lib.bias_act(x, b, r, relu=False), (x + b) + r)
    for C in (256, 288, 1024):
        x = rng.standard_normal((70, C), dtype=np.float32)
        res = rng.standard_normal((70, C), dtype=np.float32)
        g, be = rng.standard_normal(C, dtype=np.float32), rng.standard_normal(C, dtype=np.float32)
        s = (x + res).astype(np.float64)
        ref = (s - s.mean(-1, keepdims=True)) / np.sqrt(s.var(-1, keepdims=True) + 1e-5) * g + be
        np.testing.assert_allclose(emu_lib.add_layernorm(x, res, g, be), ref, atol=2e-5, rtol=1e-5)


@pytest.fixture(params=[6, 16], ids=["six_terms", "fp16_pieces"])
def terms(request):
    """The split product (include/tf_fused.h): six bf16 terms or fp16 pieces (the three-term bf16 mode was removed in round 5).
    Every case runs for the package's default product; for the other one every third case (by a hash of its name) unless TF_EMU_ALL_SCHEMES=1 -- the
    kernels are the same templates, and the emulator takes seconds per case."""
    import os
    import zlib
    from trackformer_amd import fused
    if (request.param != fused.split_terms() and os.environ.get("TF_EMU_ALL_SCHEMES") != "1"
            and zlib.crc32(request.node.nodeid.encode()) % 3):
        pytest.skip("non-default split product: every third case (TF_EMU_ALL_SCHEMES=1 runs all)")
    prev = emu_lib.set_terms(request.param)
    yield request.param
    emu_lib.set_terms(prev)


def _tol(terms):
    """Relative error of a split-product GEMM against float64: six bf16 terms and the fp16 pieces sit at fp32 round-off."""
    return 2e-6


LINEAR_SHAPES = [(200, 256, 256), (333, 256, 384), (130, 256, 1024), (130, 1024, 256), (400, 288, 288), (70, 64, 96)]


@pytest.mark.parametrize("M,K,N", LINEAR_SHAPES, ids=["%dx%dx%d" % s for s in LINEAR_SHAPES])
@pytest.mark.parametrize("relu", [False, True], ids=["plain", "relu"])
def test_split_product_linear_on_emulated_matrix_cores(M, K, N, relu, terms):
    """tf_linear_split_f32 (LDS-staged operands) and tf_linear_packed_f32 (weight fragments in MFMA order): the six-term bf16 and
    the fp16 split product on the emulated v_mfma_f32_32x32x16_bf16 / _f16 against a float64 product.  A wrong fragment layout
    anywhere gives errors of order 1; a wrong piece pairing in the six-term form errors of 2^-16."""
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    if relu:
        ref = np.maximum(ref, 0)
    if K % 32 == 0:
        y = emu_lib.linear_split(x, w, b, relu)
        assert np.abs(y - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    if K % 64 == 0:
        y2 = emu_lib.linear_packed(x, w, b, relu)
        assert np.abs(y2 - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
        if K % 32 == 0:
            assert np.array_equal(y2, y)    # same products in the same order


@pytest.mark.parametrize("M,K,N", [(4200, 64, 256), (4200, 512, 128), (4200, 64, 384), (300, 96, 200)], ids=lambda v: str(v))
def test_split_product_linear_block_shapes(M, K, N, terms):
    """The block shapes tf_linear_split_f32 picks beyond the few-rows kernels: 64 x 128 with register prefetch (many rows),
    128 x 64 (K >= 512, N <= 256), 64 x 128 without prefetch (256 < N < 512), 64 x 64 (few rows, K / 32 not one of the ring
    kernel's trip counts)."""
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    assert np.abs(emu_lib.linear_split(x, w) - ref).max() < _tol(terms) * np.abs(ref).max()


def test_six_term_product_is_fp32_accurate():
    """The claim behind the default: with (hi, mid, lo) pieces and six terms the result is as close to the exact product as an
    fp32 GEMM is (here: numpy's sgemm), and so is the fp16 product; the three bf16 pieces reconstruct the operand exactly."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((256, 1024), dtype=np.float32) * np.exp(rng.standard_normal((256, 1024)) * 2).astype(np.float32))
    w = (rng.standard_normal((256, 1024), dtype=np.float32) / 32).astype(np.float32)
    hi, mid, lo = emu_lib.bf16_split(w, 6)
    f = lambda h: (h.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    assert np.array_equal(f(hi) + f(mid) + f(lo), w.astype(np.float64))
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T     # sum |x||w|: what rounding errors scale with
    err = {}
    for t in (6, 16):
        prev = emu_lib.set_terms(t)
        try:
            err[t] = (np.abs(emu_lib.linear_split(x, w) - ref) / scale).max()
        finally:
            emu_lib.set_terms(prev)
    sgemm = (np.abs(x @ w.T - ref) / scale).max()
    print("max |err| / sum |x||w|: six terms %.2e, fp16 pieces %.2e, numpy sgemm %.2e" % (err[6], err[16], sgemm))
    assert err[6] < 2 * sgemm and err[16] < 2 * sgemm


@pytest.mark.parametrize("xs,ws", [(1.0, 0.05), (1e-3, 0.05), (300.0, 0.05), (1.0, 1e-4), (1.0, 30.0)],
                         ids=["x1", "x1e-3", "x300", "w1e-4", "w30"])
def test_fp16_product_is_fp32_class_whatever_the_magnitudes(xs, ws):
    """The claim behind the DEFAULT (include/tf_fused.h, terms = 16): two fp16 pieces per operand with the lower activation piece
    stored times 2^11 and the weights scaled per output channel are fp32-class over the magnitudes a network produces -- small and
    large activations, small and large weights, output channels whose scales differ by 1e5, rows of activations that differ by 1e4
    -- where plain fp16 pieces would fall into the subnormals (tools/experiments/f16_split.py).  Error against float64,
    normalised per output by sum |x||w|: within a factor of two of numpy's sgemm, through the packed kernel and the piece-tensor
    kernel (which must agree bit for bit)."""
    rng = np.random.default_rng(11)
    M, K, N = 150, 256, 192
    x = (rng.standard_normal((M, K)) * xs).astype(np.float32)
    w = (rng.standard_normal((N, K)) * ws).astype(np.float32)
    w[::7] *= 1e-3          # channels of very different scale (FrozenBN folded into a convolution)
    w[3::11] *= 100.0
    x[::5] *= 1e-2          # rows of very different scale
    x[2::9] *= 1e2
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T
    prev = emu_lib.set_terms(16)
    try:
        packed = emu_lib.linear_packed(x, w)
        pieces = emu_lib.linear_split(x, w)
    finally:
        emu_lib.set_terms(prev)
    assert np.array_equal(packed, pieces)
    err = float((np.abs(packed - ref) / scale).max())
    sgemm = float((np.abs((x @ w.T).astype(np.float64) - ref) / scale).max())
    # the scheme's floor: an activation below 2^-10 is represented to an ABSOLUTE 2^-32 (split_product.h), i.e. an output carries
    # up to 2^-32 sum_k |w_nk| that does not shrink with the row -- harmless next to O(1) rows, visible relative to a row of 1e-5s
    floor = 2.0 ** -31 * np.abs(w).astype(np.float64).sum(1)[None, :]
    excess = float(((np.abs(packed - ref) - floor) / scale).max())
    print("x ~ %g, w ~ %g: max |err| / sum |x||w|: fp16 product %.2e (beyond the absolute floor: %.2e), numpy sgemm %.2e" % (
        xs, ws, err, excess, sgemm))
    assert np.isfinite(packed).all() and excess < 2 * sgemm
    if xs >= 1.0:
        assert err < 2 * sgemm      # every |x| of these cases is above 2^-10 up to the rows scaled down by 1e-2 ... still above it


@pytest.mark.parametrize("mfma", [1, 2, 0], ids=["matrix_cores", "matrix_cores_lds_staged", "vector"])
@pytest.mark.parametrize("Lq,Lk,H,D,masked", [(100, 100, 8, 32, False), (57, 130, 8, 36, True), (33, 33, 4, 16, True), (40, 40, 2, 64, False),
                                              (280, 90, 8, 36, True), (17, 1, 2, 32, False),
                                              (20, 600, 2, 32, True), (18, 300, 2, 64, True), (20, 530, 1, 36, False)])
def test_query_self_attention_kernel(Lq, Lk, H, D, masked, mfma):
    """tf_mha_core_f32: the fp32 matrix-core kernel (round 5: v_mfma_f32_16x16x4_f32 for q k^T and P V) and the vector kernel it
    replaced, against float64 numpy.  (280 queries x 8 heads x 2 images = more workgroups than CUs: 64-key chunks, one buffer.)"""
    rng = np.random.default_rng(Lq + D)
    N = 2
    q = rng.standard_normal((N, Lq, H, D), dtype=np.float32)
    k = rng.standard_normal((N, Lk, H, D), dtype=np.float32)
    v = rng.standard_normal((N, Lk, H, D), dtype=np.float32)
    mask = None
    if masked:
        mask = np.zeros((N, Lk), np.uint8)
        mask[1, -7:] = 1
        mask[0, 3] = 1
    scale = 1.0 / np.sqrt(D)
    s = np.einsum("nlhd,njhd->nhlj", q.astype(np.float64), k.astype(np.float64)) * scale
    if masked:
        s = np.where(mask[:, None, None, :] != 0, -np.inf, s)
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = np.einsum("nhlj,njhd->nlhd", p, v.astype(np.float64))
    prev = emu_lib.set_options(mha_mfma=mfma)
    try:
        base = emu_lib.mha_core(q, k, v, scale, mask)
        np.testing.assert_allclose(base, ref, atol=2e-5, rtol=1e-4)
        if masked:   # a row whose keys are all masked gives zeros (both kernels), not NaN
            mask[1, :] = 1
            out = emu_lib.mha_core(q, k, v, scale, mask)
            assert np.array_equal(out[1], np.zeros_like(out[1]))
            np.testing.assert_allclose(out[0], ref[0], atol=2e-5, rtol=1e-4)
    finally:
        emu_lib.set_options(**prev)


# ---- kernels first written against this emulator (defaults since their hardware validation in round 3) --------------------------------------------------------------
@pytest.mark.parametrize("Lq,L,N", [(70, 8, 1), (29, 4, 2), (5, 3, 1), (800, 8, 1)], ids=["l8", "l4_n2", "l3_tiny", "cfg4_queries"])
def test_direct9_decoder_kernel_head_dim_36(Lq, L, N):
    """msda_fwd_f32_direct9 (9 lanes per pair, 7 pairs per wave; opt-in): plain entry (host and device shapes) and the
    fused entry with 2-d and 4-d reference points against the oracle, and bit-for-bit against msda_fwd_f32_buf."""
    shapes = ([(13, 21), (7, 11), (4, 6), (2, 3)] * 2)[:L]
    import torch
    value, shp, loc, attn, _ = _np(rand_inputs(seed=40 + Lq, N=N, M=8, D=36, Lq=Lq, P=4, shapes=shapes, loc_mode="wide"))
    ref_out = msda_oracle.msda_forward(value, shp, loc, attn)
    base = emu_lib.msda_forward(value, shp, loc, attn)
    prev = emu_lib.set_options(direct9=1)
    try:
        out = emu_lib.msda_forward(value, shp, loc, attn)
        np.testing.assert_allclose(out, ref_out, atol=1e-5, rtol=1e-4)
        np.testing.assert_allclose(emu_lib.msda_forward(value, shp, loc, attn, dshapes=True), ref_out, atol=1e-5, rtol=1e-4)
        assert np.abs(out - base).max() < 2e-6
        rng = np.random.default_rng(Lq)
        M, P = 8, 4
        qproj = rng.standard_normal((N, Lq, 3 * M * L * P), dtype=np.float32)
        for ref_dim in (2, 4):
            ref = (rng.random((N, Lq, L, ref_dim), dtype=np.float32) * 0.6 + 0.1).astype(np.float32)
            off = qproj[..., :2 * M * L * P].reshape(N, Lq, M, L, P, 2)
            logits = qproj[..., 2 * M * L * P:].reshape(N, Lq, M, L * P)
            e = np.exp(logits - logits.max(-1, keepdims=True))
            a = (e / e.sum(-1, keepdims=True)).reshape(N, Lq, M, L, P).astype(np.float32)
            if ref_dim == 2:
                floc = ref[:, :, None, :, None, :] + off / np.array(shapes, np.float32)[None, None, None, :, None, :]
            else:
                floc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
            expect = msda_oracle.msda_forward(value, shp, floc.astype(np.float32), a)
            got = emu_lib.msda_forward_fused(value, shp, ref, qproj, M, L, P)
            np.testing.assert_allclose(got, expect, atol=2e-5, rtol=1e-4)
    finally:
        emu_lib.set_options(**prev)


BWD2_CASES = [c for c in ENC_CASES if c[0] in ("pyramid_init", "pyramid_local_n2", "pyramid_uniform_fallbacks", "pyramid_border",
                                               "tiny_levels", "one_level", "coarse_first")]


@pytest.mark.parametrize("name,shapes,mode,N,D", BWD2_CASES, ids=[c[0] for c in BWD2_CASES])
def test_encoder_shape_backward_sorted2_kernel(name, shapes, mode, N, D):
    """msda_bwd_f32_sorted2 (the encoder-shape backward): tap arithmetic once per pair through the per-wave LDS exchange, the
    channel sums as four dot products + 8-lane DPP reductions, eight destination rows per wave in the row reduction."""
    value, loc, attn = encoder_inputs(shapes, mode, N=N, seed=len(name))
    shp = np.array(shapes, np.int64)
    rng = np.random.default_rng(3)
    grad_out = rng.standard_normal((N, value.shape[1], value.shape[2] * value.shape[3]), dtype=np.float32)
    if name == "pyramid_init":   # mix in points far outside their windows
        loc = loc.copy()
        loc[:, :, :, :, ::2, 0] += 12.0 / 42
        loc[:, :, :, :, ::2, 1] -= 9.0 / 25
    rv, rl, ra = msda_oracle.msda_backward(value, shp, loc, attn, grad_out)
    emu_lib.stats(reset=True)
    gv, gl, ga = emu_lib.msda_backward(value, shp, loc, attn, grad_out)
    assert emu_lib.stats()["wave_ops"] > 0
    np.testing.assert_allclose(gv, rv, atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(gl, rl, atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(ga, ra, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("M,K,N", [(300, 64, 256), (130, 256, 64), (200, 512, 128), (70, 128, 512)], ids=lambda v: str(v))
@pytest.mark.parametrize("relu", [False, True], ids=["plain", "relu"])
def test_split_product_linear_with_residual_epilogue(M, K, N, relu, terms):
    """tf_linear_split_res_f32: y = act(x . w^T + bias + residual) -- the closing 1 x 1 convolution of a ResNet bottleneck
    (FrozenBN shift as bias, identity branch as residual) at its channel counts; bit-identical to the plain kernel + add."""
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    r = rng.standard_normal((M, N), dtype=np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b + r
    if relu:
        ref = np.maximum(ref, 0)
    y = emu_lib.linear_split(x, w, b, relu, residual=r)
    assert np.abs(y - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    plain = emu_lib.linear_split(x, w, b, False) + r
    if relu:
        plain = np.maximum(plain, 0)
    assert np.array_equal(y, plain.astype(np.float32))


@pytest.mark.parametrize("M,K,N", [(400, 256, 256), (130, 256, 384), (100, 288, 96), (70, 1024, 256), (65, 1152, 64), (200, 64, 64)],
                         ids=lambda v: str(v))
def test_deep_prefetch_linear_for_few_rows(M, K, N, terms):
    """The few-rows kernel (<= 4096 rows): ring of 8 K-slices in registers, all loads of a K = 256 block in flight at once;
    K / 32 outside {8, 9, 32, 36} takes the 64 x 64 block kernel.  Same arithmetic and accumulation order as the packed
    kernel (bit-identical where that applies), against float64."""
    rng = np.random.default_rng(M + K)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    got = emu_lib.linear_split(x, w, b, True)
    ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64).T + b, 0)
    assert np.abs(got - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    if K % 64 == 0:
        assert np.array_equal(got, emu_lib.linear_packed(x, w, b, True))


def test_opt_in_kernels_do_not_depend_on_the_scheduling_order():
    """The emulator runs the waves of a workgroup (and the lanes of a wave between two wave operations) in an arbitrary
    order; HIPEMU_SHUFFLE randomises it.  A missing barrier / fence shows up as a result that depends on that order.  The
    kernels that have not seen hardware yet are re-run under a shuffled schedule in a fresh process (the whole file passes
    under HIPEMU_SHUFFLE=1 and =2 as well; this keeps the default suite short)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, HIPEMU_SHUFFLE="3")
    sel = "direct9 or deep_prefetch or residual_epilogue or fused_ffn or residual_layernorm or add_prologue or conv3x3"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", sel, "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("ks", [3, 1], ids=["3x3", "1x1_strided"])
@pytest.mark.parametrize("n,h,w,cin,cout,stride", [(1, 9, 11, 64, 64, 1), (2, 8, 6, 32, 128, 1), (1, 10, 13, 64, 160, 2), (1, 7, 7, 128, 64, 2),
                                                    (1, 1, 1, 32, 32, 1)], ids=lambda v: str(v))
def test_conv3x3_as_split_product(n, h, w, cin, cout, stride, ks, terms):
    """tf_conv3x3_split_f32: the bottlenecks' 3 x 3 convolutions (padding 1, stride 1 / 2) as an implicit GEMM on the
    emulated matrix cores (buffer loads: taps outside the image read zeros from beyond num_records), against torch's
    convolution in float64."""
    import torch
    rng = np.random.default_rng(h * w + cin)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, ks, ks, cin), dtype=np.float32) / np.sqrt(ks * ks * cin)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double().permute(0, 3, 1, 2),
                                     torch.from_numpy(b).double(), stride=stride, padding=1 if ks == 3 else 0).clamp_min(0).permute(0, 2, 3, 1).numpy()
    y = emu_lib.conv3x3_split(x, wt, b, relu=True, stride=stride)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    st = emu_lib.stats()
    assert st["divergent_ops"] == 0 and st["inactive_reads"] == 0


@pytest.mark.parametrize("n,hw,c,g", [(1, 300, 256, 32), (2, 77, 288, 32), (1, 1, 64, 8), (3, 129, 32, 4)], ids=lambda v: str(v))
def test_groupnorm_nhwc(n, hw, c, g):
    """tf_groupnorm_nhwc_f32 (statistics per image and group over HW x C / G, double accumulation) against numpy float64."""
    rng = np.random.default_rng(hw + c)
    x = (rng.standard_normal((n, hw, c), dtype=np.float32) * 3 + 1.5).astype(np.float32)
    ga, be = rng.standard_normal(c, dtype=np.float32), rng.standard_normal(c, dtype=np.float32)
    xr = x.astype(np.float64).reshape(n, hw, g, c // g)
    mean = xr.mean(axis=(1, 3), keepdims=True)
    var = xr.var(axis=(1, 3), keepdims=True)
    ref = ((xr - mean) / np.sqrt(var + 1e-5)).reshape(n, hw, c) * ga + be
    got = emu_lib.groupnorm_nhwc(x, ga, be, g)
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)
    assert np.array_equal(emu_lib.groupnorm_nhwc(x, ga, be, g, relu=True), np.maximum(got, 0))   # tf_groupnorm_relu_nhwc_f32: the same pass + ReLU


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_box_refine_fused(ref_dim):
    """tf_box_refine_f32 against the reference's formulation in float64 (incl. references at / outside [0, 1])."""
    rng = np.random.default_rng(ref_dim)
    rows = 401
    delta = rng.standard_normal((rows, 4)).astype(np.float32)
    ref = rng.random((rows, ref_dim)).astype(np.float32)
    ref[:5] = [[0.0, 1.0, -0.2, 1.3][:ref_dim]] * 5
    x = np.clip(ref.astype(np.float64), 0, 1)
    inv = np.log(np.maximum(x, 1e-5) / np.maximum(1 - x, 1e-5))
    v = delta.astype(np.float64).copy()
    v[:, :ref_dim] += inv
    np.testing.assert_allclose(emu_lib.box_refine(delta, ref), 1 / (1 + np.exp(-v)), atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("classes,clip", [(1, True), (20, True), (3, False)])
def test_postprocess_pack_fused(classes, clip):
    """tf_postprocess_pack_f32 (round 6) against the module chain it replaces in the tracker: DeformablePostProcess.forward
    (sigmoid, best class, boxes cxcywh -> xyxy scaled to the image), clip_boxes_to_image, the stacking of boxes / score / label.
    Boxes bit for bit (the same operations, each rounded on its own), labels equal (first class that attains the maximum, ties
    included), scores to an ulp of the exponential."""
    import torch
    from trackformer_amd.box_ops import clip_boxes_to_image
    from trackformer_amd.deformable_detr import DeformablePostProcess
    g = torch.Generator().manual_seed(classes)
    q, h, w = 403, 1080.0, 1920.0
    logits = torch.randn(1, q, classes, generator=g) * 3
    if classes > 1:
        logits[0, :7, 1] = logits[0, :7, 0]            # ties: the first class wins
        logits[0, 7:12] = 40.0                         # saturated sigmoids: all classes tie at 1.0
    boxes = torch.rand(1, q, 4, generator=g)
    boxes[0, :20, 2:] *= 3                             # boxes that overflow the image on every side
    res = DeformablePostProcess()({'pred_logits': logits, 'pred_boxes': boxes}, torch.tensor([[int(h), int(w)]]))[0]
    want_boxes = clip_boxes_to_image(res['boxes'], (int(h), int(w))) if clip else res['boxes']
    got = emu_lib.postprocess_pack(logits[0].numpy(), boxes[0].numpy(), h, w, clip)
    assert np.array_equal(got[:, :4], want_boxes.numpy())
    assert np.array_equal(got[:, 5].astype(np.int64), res['labels'].numpy())
    np.testing.assert_allclose(got[:, 4], res['scores'].numpy(), rtol=3e-7, atol=0)


@pytest.mark.parametrize("shape,out_size,qpi", [((3, 5, 7, 8), (10, 14), 3), ((4, 25, 42, 16), (50, 84), 2), ((2, 50, 84, 4), (100, 167), 1)])
def test_upsample_add_fused(shape, out_size, qpi):
    """tf_upsample_add_nhwc_f32 (round 6: the mask head's FPN merge) against F.interpolate(mode="nearest") + the broadcast add,
    bit for bit -- incl. 84 -> 167 columns, where the nearest index is not x / 2."""
    import torch
    g = torch.Generator().manual_seed(shape[1])
    low = torch.randn(*shape, generator=g)                                   # [N, h, w, C]
    fpn = torch.randn(shape[0] // qpi, *out_size, shape[3], generator=g)      # [B, H, W, C]
    up = torch.nn.functional.interpolate(low.permute(0, 3, 1, 2), size=out_size, mode="nearest")            # [N, C, H, W]
    want = (up.view(shape[0] // qpi, qpi, *up.shape[1:]) + fpn.permute(0, 3, 1, 2)[:, None]).flatten(0, 1)  # detr_segmentation._merge
    got = emu_lib.upsample_add(low.numpy(), fpn.numpy(), qpi)
    assert np.array_equal(got, want.permute(0, 2, 3, 1).numpy())


@pytest.mark.parametrize("lowres,pad,img,out", [((12, 20), (48, 80), (48, 80), (48, 80)), ((10, 17), (40, 67), (37, 61), (54, 96)),
                                                 ((25, 42), (100, 167), (100, 160), (67, 107))])
def test_mask_label_map_fused(lowres, pad, img, out):
    """tf_mask_label_map_f32 (round 6) against the chain it replaces in the tracker: PostProcessSegm (bilinear to the padded size,
    sigmoid, crop, nearest to the original size) -> stack -> max over the tracks -> threshold.  Same label map: pixels where the
    two best tracks are closer than fp32 round-off (the CPU's and the kernel's exponentials differ in the last bit) excepted --
    and there are none to speak of."""
    import torch
    from trackformer_amd.detr_segmentation import PostProcessSegm
    g = torch.Generator().manual_seed(lowres[0])
    n = 7
    logits = torch.randn(1, n, *lowres, generator=g) * 3
    logits[0, 5] = logits[0, 2]                        # a tie between two tracks everywhere: the first wins
    order = [3, 0, -1, 6, 2, 5, 1]                     # track i -> row of logits (-1: no mask)
    seg = PostProcessSegm()([{}], {'pred_masks': logits}, torch.tensor([list(out)]), torch.tensor([list(img)]), return_probs=True)[0]['masks'].squeeze(1)
    # PostProcessSegm resizes to the PADDED BATCH size = the largest `size` of the batch: here the one image's own (pad == img
    # unless the caller pads); the kernel takes both, so feed the module the padded size by hand for the cropped case
    if pad != img:
        cur = torch.nn.functional.interpolate(logits, size=pad, mode="bilinear", align_corners=False)[0].sigmoid()
        seg = torch.nn.functional.interpolate(cur[:, :img[0], :img[1]].unsqueeze(1), size=out, mode="nearest").squeeze(1)
    probs = torch.stack([seg[r] if r >= 0 else torch.full(out, -1.0) for r in order])
    best, owner = probs.max(dim=0)
    want = torch.where(best > 0.5, owner, torch.full_like(owner, -1)).to(torch.int16).numpy()
    got = emu_lib.mask_label_map(logits[0].numpy(), order, pad, img, out)
    assert got.shape == want.shape
    assert (got != want).mean() < 1e-3
    assert set(np.unique(got).tolist()) <= {-1, 0, 1, 3, 4, 6}     # track 2 has no mask; track 5's row equals track 4's: the first wins
    assert (got == 4).any()


@pytest.mark.parametrize("c,groups,hw", [(16, 8, (9, 37)), (32, 8, (17, 33)), (16, 4, (8, 32))])
def test_groupnorm_relu_conv_to_one_channel_fused(c, groups, hw):
    """tf_groupnorm_relu_conv3x3_c1_nhwc_f32 (round 6: the end of the mask head, out_lay(relu(gn5(x)))) against torch in float64:
    ragged tiles, the zero padding applied to the NORMALISED activation."""
    import torch
    g = torch.Generator().manual_seed(c + groups)
    n, (H, W) = 3, hw
    x = torch.randn(n, c, H, W, generator=g) * 2 + 0.5
    gn = torch.nn.GroupNorm(groups, c)
    conv = torch.nn.Conv2d(c, 1, 3, padding=1)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(c, generator=g) * 0.2)
        want = conv.double()(torch.relu(gn.double()(x.double())))[:, 0]
    got = emu_lib.groupnorm_relu_conv3x3_c1(x.permute(0, 2, 3, 1).numpy(), gn.weight.detach().float().numpy(), gn.bias.detach().float().numpy(),
                                            conv.weight.detach().float()[0].permute(1, 2, 0).reshape(9, c).numpy(), float(conv.bias.detach()), groups)
    np.testing.assert_allclose(got, want.numpy(), atol=2e-5, rtol=1e-5)


# ------------------------------------------------------------------ one-launch feed-forward block (opt-in, ffn_fused.hip)
def _ffn_case(M, F, seed, D=256):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, D), dtype=np.float32)
    w1 = (rng.standard_normal((F, D), dtype=np.float32) / 16).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(F, dtype=np.float32)).astype(np.float32)
    w2 = (rng.standard_normal((D, F), dtype=np.float32) / np.sqrt(F)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(D, dtype=np.float32)).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D, dtype=np.float32)).astype(np.float32)
    be = (0.1 * rng.standard_normal(D, dtype=np.float32)).astype(np.float32)
    return x, w1, b1, w2, b2, g, be


@pytest.mark.parametrize("M,F,ti", [(200, 1024, 3), (97, 1024, 3), (96, 256, 3), (130, 1024, 2), (33, 128, 1), (1, 1024, 3)])
def test_fused_ffn_equals_the_two_packed_linears_bit_for_bit(M, F, ti, terms):
    """tf_ffn_fused_f32 without LayerNorm: linear1 -> ReLU -> linear2 -> + residual with the hidden activation kept in LDS
    gives exactly what tf_linear_packed_f32 (relu) -> tf_linear_packed_f32 -> + residual gives (same split, same order of
    the matrix-core sums); rows behind M (the last block's tail) are never written."""
    x, w1, b1, w2, b2, _, _ = _ffn_case(M, F, M + F)
    ref = emu_lib.linear_packed(emu_lib.linear_packed(x, w1, b1, relu=True), w2, b2) + x
    prev = emu_lib.set_options(ffn_ti=ti)
    try:
        y = emu_lib.ffn_fused(x, w1, b1, w2, b2, residual=x, guard_rows=3)
    finally:
        emu_lib.set_options(**prev)
    assert np.array_equal(y[:M], ref)
    assert np.isnan(y[M:]).all()
    f64 = np.maximum(x.astype(np.float64) @ w1.T + b1, 0) @ w2.T.astype(np.float64) + b2 + x
    assert np.abs(ref - f64).max() < _tol(terms) * max(1.0, np.abs(f64).max())
    st = emu_lib.stats()
    assert st["divergent_ops"] == 0 and st["inactive_reads"] == 0


def test_fused_ffn_without_biases_and_residual(terms):
    x, w1, _, w2, _, _, _ = _ffn_case(70, 512, 5)
    ref = emu_lib.linear_packed(emu_lib.linear_packed(x, w1, None, relu=True), w2, None)
    assert np.array_equal(emu_lib.ffn_fused(x, w1, None, w2, None), ref)


@pytest.mark.parametrize("M,ti", [(200, 3), (65, 2), (40, 1)])
def test_fused_ffn_layernorm_epilogue(M, ti, terms):
    """The LayerNorm of the block (norm2 / norm3) in the epilogue: two-pass statistics over a row that is spread over two
    half-waves and four waves; against float64 and against tf_add_layernorm_f32 on the un-normalised result."""
    x, w1, b1, w2, b2, g, be = _ffn_case(M, 1024, 7 * M)
    prev = emu_lib.set_options(ffn_ti=ti)
    try:
        pre = emu_lib.ffn_fused(x, w1, b1, w2, b2, residual=x)
        y = emu_lib.ffn_fused(x, w1, b1, w2, b2, residual=x, ln=(g, be), eps=1e-5, guard_rows=2)
    finally:
        emu_lib.set_options(**prev)
    assert np.isnan(y[M:]).all()
    p64 = pre.astype(np.float64)
    ln64 = (p64 - p64.mean(1, keepdims=True)) / np.sqrt(p64.var(1, keepdims=True) + 1e-5) * g + be
    assert np.abs(y[:M] - ln64).max() < 5e-6 * max(1.0, np.abs(ln64).max())
    sep = emu_lib.add_layernorm(pre, None, g, be, 1e-5)
    assert np.abs(y[:M] - sep).max() < 5e-6


def test_fused_ffn_rejects_what_it_does_not_cover():
    x, w1, b1, w2, b2, _, _ = _ffn_case(10, 128, 1)
    with pytest.raises(RuntimeError):   # d_model != 256
        emu_lib.ffn_fused(x[:, :128].copy(), w1[:, :128].copy(), b1, w2[:128].copy(), b2[:128].copy())
    with pytest.raises(RuntimeError):   # d_ffn not a multiple of 128
        emu_lib.ffn_fused(x, w1[:64].copy(), b1[:64].copy(), w2[:, :64].copy(), b2)


@pytest.mark.parametrize("M,ti", [(200, 0), (97, 3), (130, 2), (33, 1), (1, 0)])
def test_linear_residual_layernorm_in_one_launch(M, ti, terms):
    """tf_linear_res_ln_f32 (opt-in): output projection + residual add (+ LayerNorm).  Without the norm bit-identical to
    tf_linear_packed_f32 + residual; with it equal to tf_add_layernorm_f32 on that up to rounding; guard rows untouched."""
    rng = np.random.default_rng(M)
    x = rng.standard_normal((M, 256), dtype=np.float32)
    w = (rng.standard_normal((256, 256), dtype=np.float32) / 16).astype(np.float32)
    b = rng.standard_normal(256, dtype=np.float32)
    r = rng.standard_normal((M, 256), dtype=np.float32)
    g = (1 + 0.1 * rng.standard_normal(256, dtype=np.float32)).astype(np.float32)
    be = (0.1 * rng.standard_normal(256, dtype=np.float32)).astype(np.float32)
    ref = emu_lib.linear_packed(x, w, b) + r
    prev = emu_lib.set_options(linln_ti=ti)
    try:
        y = emu_lib.linear_res_ln(x, w, b, r, guard_rows=3)
        yl = emu_lib.linear_res_ln(x, w, b, r, ln=(g, be), guard_rows=3)
        y0 = emu_lib.linear_res_ln(x, w)
    finally:
        emu_lib.set_options(**prev)
    assert np.array_equal(y[:M], ref) and np.isnan(y[M:]).all() and np.isnan(yl[M:]).all()
    assert np.array_equal(y0, emu_lib.linear_packed(x, w))
    assert np.abs(yl[:M] - emu_lib.add_layernorm(ref, None, g, be, 1e-5)).max() < 5e-6
    st = emu_lib.stats()
    assert st["divergent_ops"] == 0 and st["inactive_reads"] == 0


def test_one_launch_blocks_validate_their_arguments():
    """tf_ffn_fused_f32 / tf_linear_res_ln_f32: NULL pointers, unsupported sizes, a LayerNorm weight without its bias and
    misaligned pointers come back as status codes (tf_msda.h) before anything is launched."""
    import ctypes
    L = emu_lib.lib()
    buf = emu_lib._aligned(np.zeros((64, 256), np.float32))
    wp = emu_lib._packed(np.zeros((256, 256), np.float32))
    w1 = emu_lib._packed(np.zeros((128, 256), np.float32))
    w2 = emu_lib._packed(np.zeros((256, 128), np.float32))
    y = emu_lib._aligned(np.zeros((64, 256), np.float32))
    p, f0 = (lambda a: a.ctypes.data), ctypes.c_float(1e-5)
    NULLP, BAD = -1, -2
    lin = lambda x=p(buf), w=p(wp), b=None, r=None, g=None, be=None, out=p(y), M=64, K=256, N=256, T=emu_lib.TERMS: \
        L.tf_linear_res_ln_f32(x, w, b, r, g, be, f0, out, M, K, N, T, None)
    assert lin() == 0
    assert lin(x=None) == NULLP and lin(w=None) == NULLP and lin(out=None) == NULLP
    assert lin(g=p(buf)) == NULLP                       # LayerNorm weight without bias
    assert lin(M=0) == BAD and lin(K=320, N=320) == BAD and lin(N=128) == BAD and lin(K=288) == BAD   # square, 256 or 288
    assert lin(x=p(buf) + 4) == BAD and lin(r=p(buf) + 8) == BAD   # not 16-byte aligned
    assert lin(M=(1 << 22)) == BAD                      # 32-bit buffer offsets
    assert lin(T=4) == BAD and lin(T=0) == BAD and lin(T=3) == BAD   # terms: 6 or 16 (3 was removed in round 5)
    ffn = lambda x=p(buf), a=p(w1), c=p(w2), out=p(y), M=64, D=256, F=128, g=None, be=None, T=emu_lib.TERMS: \
        L.tf_ffn_fused_f32(x, a, None, c, None, None, g, be, f0, out, M, D, F, T, None)
    assert ffn() == 0
    assert ffn(x=None) == NULLP and ffn(a=None) == NULLP and ffn(c=None) == NULLP and ffn(out=None) == NULLP
    assert ffn(be=p(buf)) == NULLP
    assert ffn(M=-1) == BAD and ffn(D=128) == BAD and ffn(F=64) == BAD and ffn(F=200) == BAD   # F: >= one chunk, multiple of 16
    assert ffn(out=p(y) + 4) == BAD and ffn(T=5) == BAD and ffn(T=3) == BAD


@pytest.mark.parametrize("n,h,w,c", [(1, 20, 33, 64), (2, 7, 8, 8), (1, 1, 1, 4), (1, 2, 5, 12)], ids=lambda v: str(v))
def test_bias_relu_maxpool_equals_the_separate_passes(n, h, w, c):
    """tf_bias_relu_maxpool_f32 (opt-in stem route): bit-identical to relu(x + b) followed by torch's MaxPool2d(3, 2, 1)."""
    import torch
    rng = np.random.default_rng(h * w + c)
    x = rng.standard_normal((n, h, w, c), dtype=np.float32)
    b = rng.standard_normal(c, dtype=np.float32)
    ref = torch.nn.functional.max_pool2d(torch.relu(torch.from_numpy(x).permute(0, 3, 1, 2) + torch.from_numpy(b).view(1, -1, 1, 1)),
                                         3, 2, 1).permute(0, 2, 3, 1).numpy()
    got = emu_lib.bias_relu_maxpool(x, b)
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.parametrize("D,M,F,ti", [(288, 150, 1024, 2), (288, 70, 1024, 1), (288, 97, 1024, 3), (288, 40, 96, 1), (288, 33, 160, 2),
                                      (256, 50, 192, 2)])
def test_fused_ffn_hidden_288_and_ragged_hidden_widths(D, M, F, ti, terms):
    """The multi-frame models' hidden size (288: three waves x three output tiles, hidden chunks of 96) and hidden widths
    that are not a multiple of the chunk (1024 = 10 x 96 + 64: the last chunk runs over zero-padded weight columns and
    clamped W2 k-steps).  Bit-identical to the two packed linears + residual; LayerNorm epilogue against float64."""
    x, w1, b1, w2, b2, g, be = _ffn_case(M, F, D + M + F, D=D)
    # tf_linear_split_f32 (any K % 32 == 0) gives the packed kernel's bits: the reference for K = 288 / ragged widths
    ref = emu_lib.linear_split(emu_lib.linear_split(x, w1, b1, True), w2, b2) + x
    prev = emu_lib.set_options(ffn_ti=ti)
    try:
        y = emu_lib.ffn_fused(x, w1, b1, w2, b2, residual=x, guard_rows=2)
        yl = emu_lib.ffn_fused(x, w1, b1, w2, b2, residual=x, ln=(g, be), eps=1e-5, guard_rows=2)
    finally:
        emu_lib.set_options(**prev)
    assert np.array_equal(y[:M], ref) and np.isnan(y[M:]).all() and np.isnan(yl[M:]).all()
    p64 = ref.astype(np.float64)
    ln64 = (p64 - p64.mean(1, keepdims=True)) / np.sqrt(p64.var(1, keepdims=True) + 1e-5) * g + be
    assert np.abs(yl[:M] - ln64).max() < 5e-6 * max(1.0, np.abs(ln64).max())
    st = emu_lib.stats()
    assert st["divergent_ops"] == 0 and st["inactive_reads"] == 0


@pytest.mark.parametrize("M,ti", [(150, 0), (70, 1), (97, 3), (64, 2)])
def test_linear_residual_layernorm_hidden_288(M, ti, terms):
    rng = np.random.default_rng(M + 288)
    D = 288
    x = rng.standard_normal((M, D), dtype=np.float32)
    w = (rng.standard_normal((D, D), dtype=np.float32) / 17).astype(np.float32)
    b = rng.standard_normal(D, dtype=np.float32)
    r = rng.standard_normal((M, D), dtype=np.float32)
    g = (1 + 0.1 * rng.standard_normal(D, dtype=np.float32)).astype(np.float32)
    be = (0.1 * rng.standard_normal(D, dtype=np.float32)).astype(np.float32)
    f64 = x.astype(np.float64) @ w.T.astype(np.float64) + b + r
    prev = emu_lib.set_options(linln_ti=ti)
    try:
        y = emu_lib.linear_res_ln(x, w, b, r, guard_rows=3)
        yl = emu_lib.linear_res_ln(x, w, b, r, ln=(g, be), guard_rows=3)
    finally:
        emu_lib.set_options(**prev)
    ref = emu_lib.linear_split(x, w, b) + r        # K = 288 is not a multiple of 64: the unpacked kernel is the reference here
    assert np.array_equal(y[:M], ref) and np.isnan(y[M:]).all() and np.isnan(yl[M:]).all()
    assert np.abs(ref - f64).max() < 1e-4 * max(1.0, np.abs(f64).max())
    assert np.abs(yl[:M] - emu_lib.add_layernorm(ref, None, g, be, 1e-5)).max() < 5e-6


@pytest.mark.parametrize("M,K,N", [(300, 256, 384), (5000, 256, 384), (400, 256, 384), (4500, 256, 256), (130, 288, 288)])
def test_linear_with_add_prologue_is_bit_identical(M, K, N, terms):
    """tf_linear_split_add_f32: (x + pos) . w^T with the add done while the activation tile is staged equals the separate
    add followed by tf_linear_split_f32 bit for bit (the three block shapes it dispatches to)."""
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((M, K), dtype=np.float32)
    pos = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    ref = emu_lib.linear_split(x + pos, w, b)
    assert np.array_equal(emu_lib.linear_split_add(x, pos, w, b), ref)


def test_lds_bank_conflict_model_on_the_encoder_kernel():
    """HIPEMU_LDS_TRACK=1: the emulator's ds_read_b128 bank model (four fixed 16-lane groups, 64 banks, broadcast) counts the
    LDS cycles of the LDS-window kernel's gathers.  Rounds 2-5: hardware counters put 34-35 % of the encoder kernel's LDS cycles
    down to bank conflicts (profiles/r02_msda_fwd_pquad_pmc.json, r05_msda_fwd_pquad2_pmc.json) and the model agreed (6.0 cycles
    per gather, tools/lds_conflict_study.py).  Round 6: msda_fwd_f32_pquad2's gather splits a quad's lanes by tap COLUMN
    (x0 / x0 + 1: neighbouring LDS rows, opposite parities whatever the data) and rotates the piece order over the quads of an
    LDS cycle -- every gather instruction is conflict-free BY CONSTRUCTION: exactly 4 cycles, on any sampling pattern (option
    pquad_cf; on MI355X SQ_LDS_BANK_CONFLICT went 3.59 M -> 0 per launch and the launch time did not improve, so it is not the
    default: profiles/r06_msda_pquad2_conflict_free.txt).  The result must not depend on the accounting."""
    import os
    import subprocess
    import sys
    code = (
        "import os\n"
        "import numpy as np\n"
        "from tests import emu_lib\n"
        "from tests.test_emu_kernels import _fused_case, PYR\n"
        "emu_lib.set_options(pquad_cf=int(os.environ['CF']))\n"
        "value, refp, qproj, _, _ = _fused_case(PYR, 1, 32, seed=3, spread=0.8)\n"
        "out = emu_lib.msda_forward_fused(value, np.array(PYR, np.int64), refp, qproj, 8, len(PYR), 4)\n"
        "st = emu_lib.stats()\n"
        "print('RESULT', st['lds_b128_reads'], st['lds_b128_cycles'], float(np.abs(out).sum()))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for track, cf in (("1", "1"), ("0", "1"), ("1", "0")):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPEMU_LDS_TRACK=track, CF=cf), capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        res[track, cf] = [float(v) for v in r.stdout.split("RESULT")[1].split()]
    reads, cycles, checksum = res["1", "1"]
    assert reads > 1000 and cycles == 4.0 * reads, (reads, cycles)           # option pquad_cf: conflict-free by construction
    assert res["0", "1"][0] == 0 and res["0", "1"][2] == checksum
    reads, cycles, _ = res["1", "0"]                                          # the default gather: between conflict-free and two-way
    assert reads > 1000 and 4.0 < cycles / reads <= 8.0, (reads, cycles)


@pytest.mark.parametrize("n,h,w", [(1, 37, 70), (2, 16, 19), (1, 8, 300), (1, 5, 5), (1, 1, 1)], ids=lambda v: str(v))
def test_stem_convolution_as_split_product(n, h, w, terms):
    """tf_stem_conv7x7_f32: the 7 x 7 / stride 2 / padding 3 stem convolution (3 -> 64) as an implicit GEMM on the emulated
    matrix cores -- K ordered (plane, kernel row, 8 padded taps), zero patch outside the image, ragged tile edges -- against
    torch's convolution in float64, with and without the shift + ReLU epilogue."""
    import torch
    rng = np.random.default_rng(h * w)
    x = rng.standard_normal((n, 3, h, w), dtype=np.float32)
    wt = (rng.standard_normal((64, 3, 7, 7), dtype=np.float32) / 12).astype(np.float32)
    b = rng.standard_normal(64, dtype=np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), None, stride=2, padding=3)
    ref = ref.permute(0, 2, 3, 1).numpy()
    y = emu_lib.stem_conv(x, wt)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    y2 = emu_lib.stem_conv(x, wt, b, relu=True)
    assert np.abs(y2 - np.maximum(ref + b, 0)).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    st = emu_lib.stats()
    assert st["divergent_ops"] == 0 and st["inactive_reads"] == 0


@pytest.mark.parametrize("n,h,w,cin,cout,stride,ksplit", [(1, 9, 11, 256, 64, 1, 4), (2, 7, 5, 160, 128, 2, 3), (1, 6, 6, 64, 192, 1, 2)], ids=lambda v: str(v))
def test_conv1x1_split_k(n, h, w, cin, cout, stride, ksplit, terms):
    """tf_conv1x1_splitk_f32: the reducing 1 x 1 convolutions of layer3 / layer4 (few output pixels under K = 1024 / 2048) with
    the K loop cut into pieces -- against torch float64 and against the unsplit kernel (same products, another sum order)."""
    import torch
    rng = np.random.default_rng(h * w + cin + ksplit)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, 1, 1, cin), dtype=np.float32) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double().permute(0, 3, 1, 2),
                                     torch.from_numpy(b).double(), stride=stride).clamp_min(0).permute(0, 2, 3, 1).numpy()
    y = emu_lib.conv3x3_splitk(x, wt, b, relu=True, stride=stride, ksplit=ksplit)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    base = emu_lib.conv3x3_split(x, wt, b, relu=True, stride=stride)
    assert np.abs(y - base).max() < 1e-5 * max(1.0, np.abs(ref).max())
    assert np.array_equal(emu_lib.conv3x3_splitk(x, wt, b, relu=True, stride=stride, ksplit=1), base)


@pytest.mark.parametrize("n,h,w,cin,cout,stride,ksplit", [(1, 9, 11, 128, 64, 2, 4), (1, 7, 6, 256, 256, 2, 9), (2, 5, 5, 64, 128, 1, 18),
                                                          (1, 6, 7, 96, 64, 1, 5), (1, 4, 4, 32, 64, 2, 1)], ids=lambda v: str(v))
def test_conv3x3_split_k(n, h, w, cin, cout, stride, ksplit, terms):
    """tf_conv3x3_splitk_f32: the K loop of the 3 x 3 convolution cut into pieces (partial sums in a workspace, added in a
    fixed order by a second launch, then bias / ReLU) -- the extra pyramid level's 2048 -> 256 projection at 13 x 21 and
    layer4's convolutions have few output pixels under a long K.  Against torch float64 and against the unsplit kernel
    (same products; the sum order differs, so up to rounding)."""
    import torch
    rng = np.random.default_rng(h * w + cin + ksplit)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, 3, 3, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double().permute(0, 3, 1, 2),
                                     torch.from_numpy(b).double(), stride=stride, padding=1).clamp_min(0).permute(0, 2, 3, 1).numpy()
    y = emu_lib.conv3x3_splitk(x, wt, b, relu=True, stride=stride, ksplit=ksplit)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    base = emu_lib.conv3x3_split(x, wt, b, relu=True, stride=stride)
    assert np.abs(y - base).max() < 1e-5 * max(1.0, np.abs(ref).max())
    if ksplit == 1:
        assert np.array_equal(y, base)


# ------------------------------------------------------------------ the stream GEMM's round-4 forms (linear_stream.hip)
@pytest.mark.parametrize("M,K,N", [(300, 64, 256), (4200, 64, 64), (130, 256, 128), (200, 512, 64), (70, 128, 512), (333, 64, 200)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("relu", [False, True], ids=["plain", "relu"])
def test_stream_gemm_residual_epilogue_and_narrow_blocks(M, K, N, relu, terms):
    """tf_linear_packed_f32 with a residual (the closing 1 x 1 convolution of a bottleneck) and with 64- / 128-column blocks:
    the bits of tf_linear_split_res_f32 (same products in the same order), nothing written behind row M."""
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    r = rng.standard_normal((M, N), dtype=np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b + r
    if relu:
        ref = np.maximum(ref, 0)
    y = emu_lib.linear_packed(x, w, b, relu, residual=r, guard_rows=3)
    assert np.isnan(y[M:]).all()
    assert np.abs(y[:M] - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    assert np.array_equal(y[:M], emu_lib.linear_split(x, w, b, relu, residual=r))
    assert np.array_equal(emu_lib.linear_packed(x, w, b, relu), emu_lib.linear_split(x, w, b, relu))
    for ti in (1, 2, 4):   # every row-tile count of the block shape the width selects
        prev = emu_lib.set_options(linear_stream_ti=ti)
        try:
            assert np.array_equal(emu_lib.linear_packed(x, w, b, relu, residual=r), y[:M]), ti
        finally:
            emu_lib.set_options(**prev)


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
@pytest.mark.parametrize("M,K,N,relu", [(130, 64, 64, False), (300, 256, 256, True), (70, 128, 96, False), (513, 192, 384, True)],
                         ids=lambda v: str(v))
def test_lds_dma_gemm_is_bit_identical(M, K, N, relu, mode, terms):
    """The LDS-DMA GEMM behind tf_linear_packed_f32 (dma_gemm_kernel, round 6; option linear_dma = block shape): both operands
    global -> LDS by LDS-DMA into a ring of K-slices with counted s_waitcnt vmcnt, the A fragments read as fp32 from an XOR-swizzled
    image and cut into pieces in registers.  Every block shape / ring depth: the bits of tf_linear_split_res_f32 (same products, same
    order), with bias, residual and ReLU, ragged M and N, nothing written behind row M; the emulator lands a DMA only at the
    s_waitcnt that covers it, so a read in front of its wait would see poisoned LDS."""
    rng = np.random.default_rng(M + K + N + mode)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N, dtype=np.float32)
    r = rng.standard_normal((M, N), dtype=np.float32)
    prev = emu_lib.set_options(linear_dma=mode)
    try:
        y = emu_lib.linear_packed(x, w, b, relu, residual=r, guard_rows=3)
        y0 = emu_lib.linear_packed(x, w, None, relu)
    finally:
        emu_lib.set_options(**prev)
    assert np.isnan(y[M:]).all()
    assert np.array_equal(y[:M], emu_lib.linear_split(x, w, b, relu, residual=r))
    assert np.array_equal(y0, emu_lib.linear_split(x, w, None, relu))
    st = emu_lib.stats()
    assert st["divergent_ops"] == 0 and st["inactive_reads"] == 0


@pytest.mark.parametrize("n,h,w,cin,cout,stride,ks", [(1, 9, 11, 64, 64, 1, 3), (2, 8, 6, 64, 128, 1, 3), (1, 10, 13, 64, 160, 2, 3),
                                                       (1, 7, 7, 128, 64, 2, 1), (1, 12, 9, 128, 256, 1, 3), (1, 6, 5, 192, 320, 1, 1),
                                                       (1, 1, 1, 64, 64, 1, 3)], ids=lambda v: str(v))
def test_convolution_through_the_stream_gemm(n, h, w, cin, cout, stride, ks, terms):
    """tf_conv_packed_f32: 3 x 3 (padding 1) and 1 x 1 convolutions, stride 1 / 2, as an implicit GEMM with the weight fragments
    streamed and only the shifted input pixels through LDS -- the bits of the LDS-staged kernel (tf_conv3x3_split_f32), and
    torch's convolution in float64 within the split product's bound; every block shape."""
    import torch
    rng = np.random.default_rng(h * w + cin + cout)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, ks, ks, cin), dtype=np.float32) / np.sqrt(ks * ks * cin)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double().permute(0, 3, 1, 2),
                                     torch.from_numpy(b).double(), stride=stride, padding=1 if ks == 3 else 0).clamp_min(0).permute(0, 2, 3, 1).numpy()
    prev_halo = emu_lib.set_options(conv_halo=0)   # the stream form (the halo form of the stride-1 3 x 3 layers: next test)
    try:
        y = emu_lib.conv_packed(x, wt, b, relu=True, stride=stride)
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
        assert np.array_equal(y, emu_lib.conv3x3_split(x, wt, b, relu=True, stride=stride))
        for ti in (1, 2, 4):
            prev = emu_lib.set_options(linear_stream_ti=ti)
            try:
                assert np.array_equal(emu_lib.conv_packed(x, wt, b, relu=True, stride=stride), y), ti
            finally:
                emu_lib.set_options(**prev)
        st = emu_lib.stats()
        assert st["divergent_ops"] == 0 and st["inactive_reads"] == 0
    finally:
        emu_lib.set_options(**prev_halo)


@pytest.mark.parametrize("n,h,w,cin,cout,ksplit", [(1, 9, 11, 64, 64, 1), (2, 8, 6, 64, 128, 1), (1, 12, 9, 128, 256, 1), (1, 1, 1, 64, 64, 1),
                                                    (1, 17, 19, 64, 96, 1), (2, 5, 5, 128, 128, 3), (1, 16, 8, 256, 64, 4), (1, 7, 23, 128, 320, 2),
                                                    (2, 19, 13, 32, 16, 1), (1, 33, 9, 64, 32, 1), (1, 6, 10, 288, 128, 1), (1, 9, 9, 96, 24, 3)],
                         ids=lambda v: str(v))
def test_convolution_3x3_halo_form(n, h, w, cin, cout, ksplit, terms):
    """The halo form of the stride-1 3 x 3 convolution (conv3x3_halo_kernel, round 6): a block stages the halo of its patch of output
    pixels once per 32-channel slice and runs all nine taps from it.  Against torch's convolution in float64 within the split
    product's bound -- images that are not multiples of the patch, one pixel, several images, residual + ReLU, the channel loop cut
    into pieces -- and against the stream form (the same products in another fp32 order); every block shape gives the same bits
    (the order per output element does not depend on the patch), nothing is read from inactive lanes."""
    import torch
    rng = np.random.default_rng(h * w + cin + cout + ksplit)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, 3, 3, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32)
    conv = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double().permute(0, 3, 1, 2),
                                      torch.from_numpy(b).double(), stride=1, padding=1).permute(0, 2, 3, 1)
    r = rng.standard_normal(tuple(conv.shape), dtype=np.float32)
    ref = (conv + torch.from_numpy(r).double()).clamp_min(0).numpy()
    assert emu_lib.set_options(conv_halo=1)["conv_halo"] == 1   # the default
    y = emu_lib.conv_packed(x, wt, b, relu=True, stride=1, ksplit=ksplit, residual=r)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    st = emu_lib.stats()
    assert st["divergent_ops"] == 0 and st["inactive_reads"] == 0
    for ti in (1, 2, 4):
        prev = emu_lib.set_options(linear_stream_ti=ti)
        try:
            assert np.array_equal(emu_lib.conv_packed(x, wt, b, relu=True, stride=1, ksplit=ksplit, residual=r), y), ti
        finally:
            emu_lib.set_options(**prev)
    if cin % 64:
        return   # (the stream form walks pairs of slices: Cin % 64; the mask head's 288- and 32-channel layers are the halo form's alone)
    prev = emu_lib.set_options(conv_halo=0)
    try:
        stream = emu_lib.conv_packed(x, wt, b, relu=True, stride=1, ksplit=ksplit, residual=r)
    finally:
        emu_lib.set_options(**prev)
    assert np.abs(y - stream).max() <= 4e-6 * max(1.0, np.abs(ref).max())   # same products, another summation order


@pytest.mark.parametrize("n,qpi,low,out,cin,cout,norm", [(4, 2, (5, 7), (10, 14), 32, 16, True), (3, 3, (6, 21), (12, 41), 64, 32, True),
                                                          (2, 1, (4, 5), (8, 10), 128, 64, False), (2, 2, (3, 4), (6, 8), 288, 128, True)],
                         ids=lambda v: str(v))
def test_convolution_3x3_with_the_fpn_merge_in_its_fetch(n, qpi, low, out, cin, cout, norm, terms):
    """tf_conv3x3_merge_packed_f32 (round 6, the mask head's FPN levels): conv(relu(gn(low)) up-sampled (nearest) + fpn broadcast over
    the queries of an image) with the merge -- and the previous layer's GroupNorm + ReLU -- computed in the convolution's fetch.
    Against the unfused chain of the SAME library (groupnorm, upsample_add, the halo convolution): the same products of the same
    values -- bit for bit -- and against torch in float64 within the split product's bound."""
    import torch
    rng = np.random.default_rng(n * cin + cout)
    lo = (rng.standard_normal((n, *low, cin), dtype=np.float32) * 2 + 0.3).astype(np.float32)
    fpn = rng.standard_normal((n // qpi, *out, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, 3, 3, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32)
    groups = 8
    gamma = (rng.random(cin, dtype=np.float32) + 0.5).astype(np.float32)
    beta = (rng.standard_normal(cin, dtype=np.float32) * 0.2).astype(np.float32)
    got = emu_lib.conv3x3_merged(lo, fpn, qpi, wt, b, relu=False, gn=(gamma, beta, groups, 1e-5) if norm else None)
    act = emu_lib.groupnorm_nhwc(lo.reshape(n, low[0] * low[1], cin), gamma, beta, groups, relu=True).reshape(lo.shape) if norm else lo
    merged = emu_lib.upsample_add(act, fpn, qpi)
    chain = emu_lib.conv_packed(merged, wt, b, relu=False, stride=1)
    assert got.shape == chain.shape and np.array_equal(got, chain)
    t = torch.from_numpy(lo).double().permute(0, 3, 1, 2)
    if norm:
        t = torch.relu(torch.nn.functional.group_norm(t, groups, torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), 1e-5))
    t = torch.nn.functional.interpolate(t, size=out, mode="nearest")
    t = (t.view(n // qpi, qpi, *t.shape[1:]) + torch.from_numpy(fpn).double().permute(0, 3, 1, 2)[:, None]).flatten(0, 1)
    ref = torch.nn.functional.conv2d(t, torch.from_numpy(wt).double().permute(0, 3, 1, 2), torch.from_numpy(b).double(), padding=1).permute(0, 2, 3, 1).numpy()
    assert np.abs(got - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    st = emu_lib.stats()
    assert st["divergent_ops"] == 0 and st["inactive_reads"] == 0


@pytest.mark.parametrize("n,h,w,cin,cout,stride,ks,ksplit", [(1, 9, 11, 128, 64, 2, 3, 4), (1, 7, 6, 256, 256, 2, 3, 9), (2, 5, 5, 64, 128, 1, 3, 18),
                                                              (1, 9, 11, 256, 64, 1, 1, 4), (1, 6, 6, 64, 192, 1, 1, 2), (1, 4, 4, 64, 64, 2, 3, 1)],
                         ids=lambda v: str(v))
def test_stream_convolution_split_k(n, h, w, cin, cout, stride, ks, ksplit, terms):
    """tf_conv_packed_f32 with the K loop cut into pieces (partial sums in a workspace, added in a fixed order by a second launch,
    then bias / residual / ReLU): against float64 and the uncut kernel (same products, another sum order); run twice: same bits."""
    import torch
    rng = np.random.default_rng(h * w + cin + ksplit)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, ks, ks, cin), dtype=np.float32) / np.sqrt(ks * ks * cin)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32)
    pad = 1 if ks == 3 else 0
    conv = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double().permute(0, 3, 1, 2),
                                      torch.from_numpy(b).double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    r = rng.standard_normal(tuple(conv.shape), dtype=np.float32)
    ref = (conv + torch.from_numpy(r).double()).clamp_min(0).numpy()
    y = emu_lib.conv_packed(x, wt, b, relu=True, stride=stride, ksplit=ksplit, residual=r)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < _tol(terms) * max(1.0, np.abs(ref).max())
    base = emu_lib.conv_packed(x, wt, b, relu=True, stride=stride, residual=r)
    assert np.abs(y - base).max() < 1e-5 * max(1.0, np.abs(ref).max())
    assert np.array_equal(emu_lib.conv_packed(x, wt, b, relu=True, stride=stride, ksplit=ksplit, residual=r), y)


@pytest.mark.parametrize("M,ti", [(300, 2), (520, 2), (333, 0), (200, 2)])
def test_fused_blocks_tail_split_is_bit_identical(M, ti, terms):
    """The rows behind the full rounds of 64-row blocks go to a second launch of 32-row blocks (dispatch_ffn / dispatch_linln):
    a row's result does not depend on the block it is in.  The emulated device has 4 CUs (HIPEMU_CUS): a round is 256 rows, so
    300 rows = one round + 44, 520 = two rounds + 8; 200 rows: no full round, nothing to split."""
    x, w1, b1, w2, b2, g, be = _ffn_case(M, 256, 3 * M)
    outs = {}
    for split in (0, 1):
        prev = emu_lib.set_options(ffn_tail_split=split, ffn_ti=ti if ti else 3)
        try:
            outs[split] = (emu_lib.ffn_fused(x, w1, b1, w2, b2, residual=x, ln=(g, be), guard_rows=2),
                           emu_lib.linear_res_ln(x, w2[:, :256].copy(), b2, x, ln=(g, be), guard_rows=2))
        finally:
            emu_lib.set_options(**prev)
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a[:M], b[:M]) and np.isnan(b[M:]).all()


@pytest.mark.parametrize("terms", [16, 6])
def test_relu_epilogues_propagate_nan_instead_of_zeroing_it(terms):
    """an activation beyond the fp16 product's range (|x| >= 65504 * 16) makes its output row NaN; a ReLU
    epilogue written `v > 0 ? v : 0` would turn that NaN into a silent zero.  Every epilogue is `v < 0 ? 0 : v` -- torch.relu's
    behaviour: the NaN row reaches the caller (linear, packed linear, bias_act, convolution, fused feed-forward block).  With six
    bf16 terms the same input is simply in range."""
    prev_terms = emu_lib.set_terms(terms)
    try:
        rng = np.random.default_rng(3)
        M, K, N = 64, 64, 64
        x = rng.standard_normal((M, K)).astype(np.float32)
        x[5, 7] = 4.0e6          # one activation out of the fp16 product's range
        w = (rng.standard_normal((N, K)) / 8).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64).T + b, 0)
        for y in (emu_lib.linear_split(x, w, b, True), emu_lib.linear_packed(x, w, b, relu=True)):
            ok = np.ones(M, bool)
            ok[5] = False
            np.testing.assert_allclose(y[ok], ref[ok], atol=2e-5 * np.abs(ref).max())
            if terms == 16:
                assert np.isnan(y[5]).all(), "the out-of-range row must be NaN, not zeros"
            else:
                np.testing.assert_allclose(y[5], ref[5], rtol=1e-5, atol=1e-5 * np.abs(ref[5]).max())
        # a NaN that reaches the element-wise epilogue kernel stays a NaN
        v = rng.standard_normal((4, 64)).astype(np.float32)
        v[1, 3] = np.nan
        out = emu_lib.bias_act(v, b, None, relu=True)
        assert np.isnan(out[1, 3]) and np.isfinite(np.delete(out.ravel(), 64 + 3)).all()
    finally:
        emu_lib.set_terms(prev_terms)
