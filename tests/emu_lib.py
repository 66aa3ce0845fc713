"""TEST INFRASTRUCTURE: numpy front-end over libtf_msda_emu.so -- the repo's HIP kernel sources compiled for the host and
run under the SIMT emulator of tests/emu/hipemu/ (built by tests/emu/build_emu.py).  Same C ABI as libtf_msda.so
(include/tf_msda.h, include/tf_fused.h) with host pointers in place of device pointers.  Used by tests/test_emu_*.py
only; the product (trackformer_amd/) never loads it."""
import ctypes

import numpy as np

from tests.emu import build_emu

_lib = None

STAT_NAMES = ("wave_ops", "barriers", "divergent_ops", "inactive_reads", "lds_dma_bytes", "blocks", "switches",
              "lds_b128_reads", "lds_b128_cycles")


def available():
    return build_emu.find_compiler() is not None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    import os
    os.environ.setdefault("HIPEMU_POISON", "1")   # dynamic LDS starts as 0xFF bytes in every workgroup: nothing may rely on zeros
    so = build_emu.build()
    if so is None:
        raise RuntimeError("no host clang++ to build the emulated library with")
    L = ctypes.CDLL(so)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.tf_msda_set_tiled.restype = ci
    L.tf_msda_set_tiled.argtypes = [ci]
    L.tf_msda_set_option.restype = ci
    L.tf_msda_set_option.argtypes = [ctypes.c_char_p, ci]
    for suf in ("f32", "f64"):
        for tail in ("", "_dshapes"):
            f = getattr(L, "tf_msda_forward_%s%s" % (suf, tail))
            f.restype = ci
            f.argtypes = [vp] * 5 + [ci] * 7 + [vp]
            b = getattr(L, "tf_msda_backward_%s%s" % (suf, tail))
            b.restype = ci
            b.argtypes = [vp] * 8 + [ci] * 7 + [vp]
    L.tf_msda_forward_fused_f32.restype = ci
    L.tf_msda_forward_fused_f32.argtypes = [vp, vp, vp, ci, vp, ci, ci, ci, vp] + [ci] * 7 + [vp]
    L.tf_bias_act_f32.restype = ci
    L.tf_bias_act_f32.argtypes = [vp, vp, vp, ctypes.c_int64, ci, ci, vp]
    L.tf_add_layernorm_f32.restype = ci
    L.tf_add_layernorm_f32.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int64, ci, ctypes.c_float, vp]
    L.tf_stem_conv7x7_f32.restype = ci
    L.tf_stem_conv7x7_f32.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    L.tf_bias_relu_maxpool_f32.restype = ci
    L.tf_bias_relu_maxpool_f32.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
    L.tf_box_refine_f32.restype = ci
    L.tf_box_refine_f32.argtypes = [vp, vp, vp, ctypes.c_int64, ci, ctypes.c_float, vp]
    L.tf_postprocess_pack_f32.restype = ci
    L.tf_postprocess_pack_f32.argtypes = [vp, vp, vp, ctypes.c_int64, ci, ctypes.c_float, ctypes.c_float, ci, vp]
    L.tf_groupnorm_stats_nhwc_f32.restype = ci
    L.tf_groupnorm_stats_nhwc_f32.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_int64, vp]
    L.tf_conv3x3_merge_packed_f32.restype = ci
    L.tf_conv3x3_merge_packed_f32.argtypes = [vp, vp, vp, vp, vp, ci, ctypes.c_float, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.tf_mask_label_map_f32.restype = ci
    L.tf_mask_label_map_f32.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ctypes.c_float, vp]
    L.tf_upsample_add_nhwc_f32.restype = ci
    L.tf_upsample_add_nhwc_f32.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
    L.tf_groupnorm_relu_conv3x3_c1_nhwc_f32.restype = ci
    L.tf_groupnorm_relu_conv3x3_c1_nhwc_f32.argtypes = [vp, vp, vp, vp, ctypes.c_float, vp, vp, ci, ci, ci, ci, ci, ctypes.c_float, vp]
    for fn in (L.tf_groupnorm_nhwc_f32, L.tf_groupnorm_relu_nhwc_f32):
        fn.restype = ci
        fn.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ctypes.c_float, ctypes.c_int64, ctypes.c_int64, vp]
    L.tf_linear_res_ln_f32.restype = ci
    L.tf_linear_res_ln_f32.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_float, vp, ctypes.c_int64, ci, ci, ci, vp]
    L.tf_ffn_fused_f32.restype = ci
    L.tf_ffn_fused_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, vp, ctypes.c_int64, ci, ci, ci, vp]
    L.tf_linear_split_add_f32.restype = ci
    L.tf_linear_split_add_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, ci, ci, vp]
    L.tf_linear_split_f32.restype = ci
    L.tf_linear_split_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, ci, ci, ci, vp]
    L.tf_linear_split_res_f32.restype = ci
    L.tf_linear_split_res_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, ci, ci, ci, vp]
    L.tf_conv3x3_splitk_f32.restype = ci
    L.tf_conv3x3_splitk_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.tf_conv3x3_split_f32.restype = ci
    L.tf_conv3x3_split_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp] + [ci] * 7 + [vp]
    L.tf_conv1x1_splitk_f32.restype = ci
    L.tf_conv1x1_splitk_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.tf_conv1x1_strided_split_f32.restype = ci
    L.tf_conv1x1_strided_split_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp] + [ci] * 7 + [vp]
    L.tf_linear_packed_bytes.restype = ctypes.c_int64
    L.tf_linear_packed_bytes.argtypes = [ci, ci, ci]
    L.tf_linear_pack_weight_f32.restype = ci
    L.tf_linear_pack_weight_f32.argtypes = [vp, vp, ci, ci, ci, vp]
    L.tf_linear_packed_f32.restype = ci
    L.tf_linear_packed_f32.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int64, ci, ci, ci, ci, vp]
    L.tf_conv_packed_f32.restype = ci
    L.tf_conv_packed_f32.argtypes = [vp, vp, vp, vp, vp, vp] + [ci] * 10 + [vp]
    L.tf_mha_core_f32.restype = ci
    L.tf_mha_core_f32.argtypes = [vp, vp, vp, vp, vp] + [ci] * 9 + [ctypes.c_float, vp]
    L.hipemu_get_stats.restype = None
    L.hipemu_get_stats.argtypes = [vp]
    L.hipemu_reset_stats.restype = None
    L.hipemu_num_cus.restype = ci
    _lib = L
    return L


def stats(reset=False):
    buf = (ctypes.c_uint64 * len(STAT_NAMES))()
    lib().hipemu_get_stats(buf)
    if reset:
        lib().hipemu_reset_stats()
    return dict(zip(STAT_NAMES, (int(v) for v in buf)))


def set_options(**opts):
    """Sets tf_msda_set_option knobs; returns the previous values (pass them back to restore)."""
    L = lib()
    return {k: L.tf_msda_set_option(k.encode(), int(v)) for k, v in opts.items()}


def _p(a):
    return a.ctypes.data if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def msda_forward(value, shapes, loc, attn, dshapes=False):
    dt = value.dtype
    suf = "f32" if dt == np.float32 else "f64"
    value, loc, attn = _c(value, dt), _c(loc, dt), _c(attn, dt)
    shapes = _c(shapes, np.int64)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.full((N, Lq, M * D), np.nan, dt)
    fn = getattr(lib(), "tf_msda_forward_%s%s" % (suf, "_dshapes" if dshapes else ""))
    rc = fn(_p(value), _p(shapes), _p(loc), _p(attn), _p(out), N, S, M, D, L, Lq, P, None)
    if rc != 0:
        raise RuntimeError("tf_msda_forward_%s: status %d" % (suf, rc))
    return out


def msda_forward_fused(value, shapes, ref, qproj, M, L, P, off_col=0, logit_col=None):
    value, ref, qproj = _c(value, np.float32), _c(ref, np.float32), _c(qproj, np.float32)
    shapes = _c(shapes, np.int64)
    N, S, _, D = value.shape
    Lq = ref.shape[1]
    ld = qproj.shape[-1]
    if logit_col is None:
        logit_col = 2 * M * L * P
    out = np.full((N, Lq, M * D), np.nan, np.float32)
    rc = lib().tf_msda_forward_fused_f32(_p(value), _p(shapes), _p(ref), ref.shape[-1], _p(qproj), ld, off_col, logit_col,
                                         _p(out), N, S, M, D, L, Lq, P, None)
    if rc != 0:
        raise RuntimeError("tf_msda_forward_fused_f32: status %d" % rc)
    return out


def msda_backward(value, shapes, loc, attn, grad_out, dshapes=False):
    dt = value.dtype
    suf = "f32" if dt == np.float32 else "f64"
    value, loc, attn, grad_out = _c(value, dt), _c(loc, dt), _c(attn, dt), _c(grad_out, dt)
    shapes = _c(shapes, np.int64)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    gv = np.full(value.shape, np.nan, dt)
    gl = np.full(loc.shape, np.nan, dt)
    ga = np.full(attn.shape, np.nan, dt)
    fn = getattr(lib(), "tf_msda_backward_%s%s" % (suf, "_dshapes" if dshapes else ""))
    rc = fn(_p(value), _p(shapes), _p(loc), _p(attn), _p(grad_out), _p(gv), _p(gl), _p(ga), N, S, M, D, L, Lq, P, None)
    if rc != 0:
        raise RuntimeError("tf_msda_backward_%s: status %d" % (suf, rc))
    return gv, gl, ga


TERMS = 6   # terms per split product of the wrappers below (include/tf_fused.h: 6 bf16 terms, or 16 = fp16 pieces)


def set_terms(n):
    """6 (bf16 pieces) or 16 (fp16 pieces, three terms); returns the previous value."""
    global TERMS
    assert n in (6, 16)
    prev, TERMS = TERMS, n
    return prev


def bf16_split(w, terms=None):
    """w (fp32) -> (hi, mid, lo) as uint16 bit patterns of bf16, round to nearest even at every step (what fused.py hands the
    kernel)."""
    def to_bf16(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
        return r

    def to_f32(h):
        return (h.astype(np.uint32) << 16).view(np.float32)
    w = w.astype(np.float32)
    hi = to_bf16(w)
    r = w - to_f32(hi)
    mid = to_bf16(r)
    lo = to_bf16(r - to_f32(mid))
    return hi, mid, lo


def f16_split(w):
    """w [N, K] (fp32) -> (wh, wl, None, r): the fp16 scheme of include/tf_fused.h as uint16 bit patterns + the channels' factors
    r_n = 16 / t_n (what fused._split_weight hands the kernels for split_terms() == 16)."""
    w = w.astype(np.float32)
    amax = np.abs(w).max(axis=1)
    _, e = np.frexp(amax)                       # amax = m 2^e, m in [0.5, 1)
    t = np.ldexp(np.float32(1), np.clip(14 - e, -100, 100)).astype(np.float32)
    t = np.where((amax > 0) & (amax < 3.0e38), t, np.float32(1)).astype(np.float32)
    ws = (w * t[:, None]).astype(np.float32)
    hi = ws.astype(np.float16)
    lo = (ws - hi.astype(np.float32)).astype(np.float16)
    return hi.view(np.uint16), lo.view(np.uint16), None, (np.float32(16) / t).astype(np.float32)


def _pieces(w2d, terms=None):
    """16-byte aligned contiguous piece arrays of a [N, K] weight -> (p0, p1, p2 | None, scale | None)."""
    if (terms or TERMS) == 16:
        return tuple(None if p is None else _aligned16(np.ascontiguousarray(p)) for p in f16_split(w2d))
    return tuple(None if p is None else _aligned16(np.ascontiguousarray(p)) for p in bf16_split(w2d, terms)) + (None,)


def linear_split(x, w, bias=None, relu=False, residual=None):
    x, w = _c(x, np.float32), _c(w, np.float32)
    M, K = x.shape
    N = w.shape[0]
    hi, mid, lo, sc = _pieces(w)
    b = _c(bias, np.float32) if bias is not None else None
    y = np.full((M, N), np.nan, np.float32)
    if residual is not None:
        r = _c(residual, np.float32)
        rc = lib().tf_linear_split_res_f32(_p(x), _p(hi), _p(mid), _p(lo), _p(sc), _p(b), _p(r), _p(y), M, K, N, int(relu), None)
    else:
        rc = lib().tf_linear_split_f32(_p(x), _p(hi), _p(mid), _p(lo), _p(sc), _p(b), _p(y), M, K, N, int(relu), None)
    if rc != 0:
        raise RuntimeError("tf_linear_split_f32: status %d" % rc)
    return y


def linear_split_add(x, x2, w, bias=None):
    """tf_linear_split_add_f32: (x + x2) @ w^T + bias."""
    x, x2, w = _aligned(x), _aligned(x2), _c(w, np.float32)
    M, K = x.shape
    N = w.shape[0]
    hi, mid, lo, sc = _pieces(w)
    b = _aligned(bias)
    y = np.full((M, N), np.nan, np.float32)
    rc = lib().tf_linear_split_add_f32(_p(x), _p(x2), _p(hi), _p(mid), _p(lo), _p(sc), _p(b), _p(y), M, K, N, None)
    if rc != 0:
        raise RuntimeError("tf_linear_split_add_f32: status %d" % rc)
    return y


def _aligned16(a):
    """A 16-byte aligned copy of a contiguous array of any dtype."""
    buf = np.zeros(a.nbytes + 16, np.uint8)
    off = (-buf.ctypes.data) % 16
    out = buf[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def linear_packed(x, w, bias=None, relu=False, residual=None, guard_rows=0):
    x, w = _aligned(x), _c(w, np.float32)
    M, K = x.shape
    N = w.shape[0]
    pk = _packed(w)
    b = _c(bias, np.float32) if bias is not None else None
    r = _aligned(residual)
    y = np.full((M + guard_rows, N), np.nan, np.float32)
    rc = lib().tf_linear_packed_f32(_p(x), pk.ctypes.data, _p(b), _p(r), _p(y), M, K, N, int(relu), TERMS, None)
    if rc != 0:
        raise RuntimeError("tf_linear_packed_f32: status %d" % rc)
    return y


def _packed(w):
    w = _c(w, np.float32)
    N, K = w.shape
    nbytes = lib().tf_linear_packed_bytes(K, N, TERMS)
    if nbytes < 0:
        raise RuntimeError("tf_linear_packed_bytes(%d, %d) < 0" % (K, N))
    buf = np.zeros(nbytes + 16, np.uint8)
    off = (-buf.ctypes.data) % 16
    pk = buf[off:off + nbytes]
    rc = lib().tf_linear_pack_weight_f32(_p(w), pk.ctypes.data, K, N, TERMS, None)
    if rc != 0:
        raise RuntimeError("tf_linear_pack_weight_f32: status %d" % rc)
    return pk


def _aligned(a):
    """A 16-byte aligned contiguous fp32 copy (numpy only guarantees that for larger arrays)."""
    if a is None:
        return None
    a = _c(a, np.float32)
    buf = np.zeros(a.size * 4 + 16, np.uint8)
    off = (-buf.ctypes.data) % 16
    out = buf[off:off + a.size * 4].view(np.float32).reshape(a.shape)
    out[...] = a
    return out


def ffn_fused(x, w1, b1, w2, b2, residual=None, ln=None, eps=1e-5, guard_rows=0):
    """tf_ffn_fused_f32: x [M, 256], w1 [F, 256], w2 [256, F]; ln = (weight, bias) or None.  guard_rows extra rows of NaN
    behind y are returned too (nothing may be written there)."""
    x = _aligned(x)
    M, D = x.shape
    F = w1.shape[0]
    p1, p2 = _packed(w1), _packed(w2)
    b1, b2, r = _aligned(b1), _aligned(b2), _aligned(residual)
    g, be = (_aligned(ln[0]), _aligned(ln[1])) if ln is not None else (None, None)
    y = _aligned(np.full((M + guard_rows, D), np.nan, np.float32))
    rc = lib().tf_ffn_fused_f32(_p(x), p1.ctypes.data, _p(b1), p2.ctypes.data, _p(b2), _p(r), _p(g), _p(be),
                                ctypes.c_float(eps), _p(y), M, D, F, TERMS, None)
    if rc != 0:
        raise RuntimeError("tf_ffn_fused_f32: status %d" % rc)
    return y


def linear_res_ln(x, w, bias=None, residual=None, ln=None, eps=1e-5, guard_rows=0):
    """tf_linear_res_ln_f32: x [M, 256], w [256, 256]; ln = (weight, bias) or None."""
    x = _aligned(x)
    M, K = x.shape
    pk = _packed(w)
    b, r = _aligned(bias), _aligned(residual)
    g, be = (_aligned(ln[0]), _aligned(ln[1])) if ln is not None else (None, None)
    y = _aligned(np.full((M + guard_rows, w.shape[0]), np.nan, np.float32))
    rc = lib().tf_linear_res_ln_f32(_p(x), pk.ctypes.data, _p(b), _p(r), _p(g), _p(be), ctypes.c_float(eps), _p(y), M, K,
                                    w.shape[0], TERMS, None)
    if rc != 0:
        raise RuntimeError("tf_linear_res_ln_f32: status %d" % rc)
    return y


def stem_weight_matrix(w):
    """[64, 3, 7, 7] -> the [64, 176] matrix of tf_stem_conv7x7_f32: k = (c * 7 + ky) * 8 + kx, zero padded."""
    w = _c(w, np.float32)
    w2 = np.zeros((w.shape[0], 176), np.float32)
    w2[:, :168] = np.pad(w, ((0, 0), (0, 0), (0, 0), (0, 1))).reshape(w.shape[0], 168)
    return w2


def stem_conv(x_nchw, w, bias=None, relu=False):
    """tf_stem_conv7x7_f32: x [N, 3, H, W], w [64, 3, 7, 7] -> y [N, Ho, Wo, 64] (channels_last)."""
    x = _c(x_nchw, np.float32)
    N, _, H, W = x.shape
    pk = _packed(stem_weight_matrix(w))
    b = _aligned(bias)
    y = _aligned(np.full((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64), np.nan, np.float32))
    rc = lib().tf_stem_conv7x7_f32(_p(x), pk.ctypes.data, _p(b), _p(y), N, H, W, int(relu), TERMS, None)
    if rc != 0:
        raise RuntimeError("tf_stem_conv7x7_f32: status %d" % rc)
    return y


def bias_relu_maxpool(x_nhwc, bias):
    """tf_bias_relu_maxpool_f32: x [N, H, W, C] -> [N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C]."""
    x, b = _aligned(x_nhwc), _aligned(bias)
    N, H, W, C = x.shape
    out = _aligned(np.full((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), np.nan, np.float32))
    rc = lib().tf_bias_relu_maxpool_f32(_p(x), _p(b), _p(out), N, H, W, C, None)
    if rc != 0:
        raise RuntimeError("tf_bias_relu_maxpool_f32: status %d" % rc)
    return out


def mha_core(q, k, v, scale, key_mask=None):
    """q [N, Lq, H, D], k / v [N, Lk, H, D] -> out [N, Lq, H, D]."""
    q, k, v = _c(q, np.float32), _c(k, np.float32), _c(v, np.float32)
    N, Lq, H, D = q.shape
    Lk = k.shape[1]
    out = np.full(q.shape, np.nan, np.float32)
    km = np.ascontiguousarray(key_mask, dtype=np.uint8) if key_mask is not None else None
    rc = lib().tf_mha_core_f32(_p(q), _p(k), _p(v), _p(out), _p(km), N, Lq, Lk, H, D, H * D, H * D, H * D, H * D,
                               ctypes.c_float(scale), None)
    if rc != 0:
        raise RuntimeError("tf_mha_core_f32: status %d" % rc)
    return out


def bias_act(x, bias, residual=None, relu=True):
    x = _c(x, np.float32).copy()
    bias = _c(bias, np.float32)
    r = _c(residual, np.float32) if residual is not None else None
    rc = lib().tf_bias_act_f32(_p(x), _p(bias), _p(r), x.size, bias.size, int(relu), None)
    if rc != 0:
        raise RuntimeError("tf_bias_act_f32: status %d" % rc)
    return x


def add_layernorm(x, res, gamma, beta, eps=1e-5):
    x = _c(x, np.float32)
    r = _c(res, np.float32) if res is not None else None
    gamma, beta = _c(gamma, np.float32), _c(beta, np.float32)
    rows, C = x.shape
    out = np.full(x.shape, np.nan, np.float32)
    rc = lib().tf_add_layernorm_f32(_p(x), _p(r), _p(gamma), _p(beta), _p(out), rows, C, ctypes.c_float(eps), None)
    if rc != 0:
        raise RuntimeError("tf_add_layernorm_f32: status %d" % rc)
    return out


def conv3x3_split(x_nhwc, w_ohwi, bias=None, relu=False, stride=1):
    """x [N, H, W, Cin] fp32, w [Cout, 3, 3, Cin] fp32 (padding 1) or [Cout, 1, 1, Cin] (no padding) -> y [N, Hout, Wout, Cout]."""
    x, w = _c(x_nhwc, np.float32), _c(w_ohwi, np.float32)
    n, h, wd, cin = x.shape
    cout, ks = w.shape[0], w.shape[1]
    x = _aligned(x)
    hi, mid, lo, sc = _pieces(w.reshape(cout, ks * ks * cin))
    b = _c(bias, np.float32) if bias is not None else None
    pad = 1 if ks == 3 else 0
    ho, wo = (h + 2 * pad - ks) // stride + 1, (wd + 2 * pad - ks) // stride + 1
    y = np.full((n, ho, wo, cout), np.nan, np.float32)
    fn = lib().tf_conv3x3_split_f32 if ks == 3 else lib().tf_conv1x1_strided_split_f32
    rc = fn(_p(x), _p(hi), _p(mid), _p(lo), _p(sc), _p(b), _p(y), n, h, wd, cin, cout, stride, int(relu), None)
    if rc != 0:
        raise RuntimeError("tf_conv3x3_split_f32: status %d" % rc)
    return y


def conv_packed(x_nhwc, w_ohwi, bias=None, relu=False, stride=1, ksplit=1, residual=None):
    """tf_conv_packed_f32: x [N, H, W, Cin], w [Cout, ks, ks, Cin] (ks = 3: padding 1; ks = 1: none) -> y [N, Hout, Wout, Cout]."""
    x, w = _aligned(x_nhwc), _c(w_ohwi, np.float32)
    n, h, wd, cin = x.shape
    cout, ks = w.shape[0], w.shape[1]
    pk = _packed(w.reshape(cout, ks * ks * cin))
    b = _aligned(bias)
    pad = 1 if ks == 3 else 0
    ho, wo = (h + 2 * pad - ks) // stride + 1, (wd + 2 * pad - ks) // stride + 1
    y = _aligned(np.full((n, ho, wo, cout), np.nan, np.float32))
    r = _aligned(residual)
    ws = _aligned(np.full((max(ksplit, 1), n * ho * wo * cout), np.nan, np.float32)) if ksplit > 1 else None
    rc = lib().tf_conv_packed_f32(_p(x), pk.ctypes.data, _p(b), _p(r), _p(y), _p(ws), ksplit, n, h, wd, cin, cout, ks, stride,
                                  int(relu), TERMS, None)
    if rc != 0:
        raise RuntimeError("tf_conv_packed_f32: status %d" % rc)
    return y


def conv3x3_merged(low_nhwc, fpn_nhwc, q_per_image, w_ohwi, bias=None, relu=False, gn=None):
    """tf_conv3x3_merge_packed_f32: low [N, h, w, Cin], fpn [N / q_per_image, H, W, Cin], w [Cout, 3, 3, Cin] -> y [N, H, W, Cout];
    gn = (gamma, beta, groups, eps): relu(GroupNorm(low)) is applied in the fetch, from the statistics pass's raw sums."""
    low, fpn, w = _aligned(low_nhwc), _aligned(fpn_nhwc), _c(w_ohwi, np.float32)
    n, lh, lw, cin = low.shape
    _, H, W, _ = fpn.shape
    cout = w.shape[0]
    pk = _packed(w.reshape(cout, 9 * cin))
    b = _aligned(bias)
    y = _aligned(np.full((n, H, W, cout), np.nan, np.float32))
    ws = gamma = beta = None
    groups, eps = 1, 0.0
    if gn is not None:
        gamma, beta, groups, eps = _aligned(gn[0]), _aligned(gn[1]), int(gn[2]), float(gn[3])
        ws = np.empty(2 * n * groups, dtype=np.float64)
        rc = lib().tf_groupnorm_stats_nhwc_f32(_p(low), _p(ws), n, lh * lw, cin, groups, lh * lw * cin, None)
        if rc != 0:
            raise RuntimeError("tf_groupnorm_stats_nhwc_f32: status %d" % rc)
    rc = lib().tf_conv3x3_merge_packed_f32(_p(low), _p(fpn), _p(ws), _p(gamma), _p(beta), groups, ctypes.c_float(eps), pk.ctypes.data, _p(b),
                                           _p(y), n, q_per_image, lh, lw, H, W, cin, cout, int(relu), TERMS, None)
    if rc != 0:
        raise RuntimeError("tf_conv3x3_merge_packed_f32: status %d" % rc)
    return y


def conv3x3_splitk(x_nhwc, w_ohwi, bias=None, relu=False, stride=1, ksplit=4):
    """tf_conv3x3_splitk_f32 / tf_conv1x1_splitk_f32: as conv3x3_split (3 x 3 with padding 1, or 1 x 1 without: by the
    weight's shape) with the K loop cut into `ksplit` pieces."""
    x, w = _aligned(x_nhwc), _c(w_ohwi, np.float32)
    n, h, wd, cin = x.shape
    cout, ks = w.shape[0], w.shape[1]
    hi, mid, lo, sc = _pieces(w.reshape(cout, ks * ks * cin))
    b = _aligned(bias)
    pad = 1 if ks == 3 else 0
    ho, wo = (h + 2 * pad - ks) // stride + 1, (wd + 2 * pad - ks) // stride + 1
    y = _aligned(np.full((n, ho, wo, cout), np.nan, np.float32))
    ws = _aligned(np.full((max(ksplit, 1), n * ho * wo * cout), np.nan, np.float32))
    fn = lib().tf_conv3x3_splitk_f32 if ks == 3 else lib().tf_conv1x1_splitk_f32
    rc = fn(_p(x), _p(hi), _p(mid), _p(lo), _p(sc), _p(b), _p(y), _p(ws), ksplit, n, h, wd, cin, cout, stride, int(relu), None)
    if rc != 0:
        raise RuntimeError("tf_conv%dx%d_splitk_f32: status %d" % (ks, ks, rc))
    return y


def groupnorm_nhwc(x, gamma, beta, G, eps=1e-5, relu=False):
    """x [N, HW, C] -> GroupNorm over (HW, C / G) per image and group [+ ReLU: tf_groupnorm_relu_nhwc_f32]."""
    x, gamma, beta = _c(x, np.float32), _c(gamma, np.float32), _c(beta, np.float32)
    n, hw, c = x.shape
    out = np.full(x.shape, np.nan, np.float32)
    ws = np.full(2 * n * G, np.nan, np.float64)
    fn = lib().tf_groupnorm_relu_nhwc_f32 if relu else lib().tf_groupnorm_nhwc_f32
    rc = fn(_p(x), _p(gamma), _p(beta), _p(out), _p(ws), n, hw, c, G, ctypes.c_float(eps), hw * c, hw * c, None)
    if rc != 0:
        raise RuntimeError("tf_groupnorm_nhwc_f32: status %d" % rc)
    return out


def postprocess_pack(logits, boxes, img_h, img_w, clip=True):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    out = np.empty((logits.shape[0], 6), dtype=np.float32)
    rc = lib().tf_postprocess_pack_f32(_p(logits), _p(boxes), _p(out), logits.shape[0], logits.shape[1], ctypes.c_float(img_h),
                                       ctypes.c_float(img_w), 1 if clip else 0, None)
    if rc != 0:
        raise RuntimeError("tf_postprocess_pack_f32: status %d" % rc)
    return out


def mask_label_map(logits, order, pad, img, out, threshold=0.5):
    """logits [n, h, w], order [n_tracks] (row of logits or -1) -> int16 [out_h, out_w]."""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    order = np.ascontiguousarray(order, dtype=np.int32)
    n, h, w = logits.shape
    label = np.empty(tuple(out), dtype=np.int16)
    rc = lib().tf_mask_label_map_f32(_p(logits), _p(order), _p(label), len(order), h, w, pad[0], pad[1], img[0], img[1], out[0], out[1],
                                     ctypes.c_float(threshold), None)
    if rc != 0:
        raise RuntimeError("tf_mask_label_map_f32: status %d" % rc)
    return label


def upsample_add(low, fpn, q_per_image):
    """low [N, h, w, C], fpn [N / q_per_image, H, W, C] (channels innermost) -> [N, H, W, C]."""
    low = np.ascontiguousarray(low, dtype=np.float32)
    fpn = np.ascontiguousarray(fpn, dtype=np.float32)
    n, h, w, c = low.shape
    _, H, W, _ = fpn.shape
    out = np.empty((n, H, W, c), dtype=np.float32)
    rc = lib().tf_upsample_add_nhwc_f32(_p(low), _p(fpn), _p(out), n, q_per_image, h, w, H, W, c, None)
    if rc != 0:
        raise RuntimeError("tf_upsample_add_nhwc_f32: status %d" % rc)
    return out


def groupnorm_relu_conv3x3_c1(x, gamma, beta, weight, bias, groups, eps=1e-5):
    """x [N, H, W, C], weight [9, C] tap-major -> [N, H, W]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    gamma, beta = np.ascontiguousarray(gamma, dtype=np.float32), np.ascontiguousarray(beta, dtype=np.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    n, H, W, c = x.shape
    out = np.empty((n, H, W), dtype=np.float32)
    ws = np.empty(2 * n * groups, dtype=np.float64)
    rc = lib().tf_groupnorm_relu_conv3x3_c1_nhwc_f32(_p(x), _p(gamma), _p(beta), _p(weight), ctypes.c_float(bias), _p(out), _p(ws), n, H, W,
                                                     c, groups, ctypes.c_float(eps), None)
    if rc != 0:
        raise RuntimeError("tf_groupnorm_relu_conv3x3_c1_nhwc_f32: status %d" % rc)
    return out


def box_refine(delta, ref, eps=1e-5):
    delta, ref = _c(delta, np.float32), _c(ref, np.float32)
    rows = delta.shape[0]
    out = np.full(delta.shape, np.nan, np.float32)
    rc = lib().tf_box_refine_f32(_p(delta), _p(ref), _p(out), rows, ref.shape[1], ctypes.c_float(eps), None)
    if rc != 0:
        raise RuntimeError("tf_box_refine_f32: status %d" % rc)
    return out
