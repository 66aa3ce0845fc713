"""Python front-end of the fused element-wise HIP kernels (include/tf_fused.h).

Used by the inference path only (eval mode, gradients disabled) on GPU tensors; everywhere else the
callers keep the plain PyTorch formulation (these helpers return None when they do not apply)."""
import os
import threading

import torch
import torch.nn.functional as F

from . import _cabi


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _param_ok(t, x):
    """A parameter tensor the kernels can read next to `x`: same device, fp32, contiguous, 16-byte aligned."""
    return (t.device == x.device and t.dtype == torch.float32 and t.is_contiguous()
            and t.data_ptr() % 16 == 0)


def bias_act_(x, bias, residual=None, relu=True):
    """In place: x = act(x + bias[c] (+ residual)) for a channels_last [N,C,H,W] fp32 GPU tensor.
    Returns x, or None when the tensors do not qualify (caller falls back to ATen ops)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)):
        return None
    C = x.shape[1]
    if C % 4 or not _param_ok(bias, x) or x.data_ptr() % 16:
        return None
    if residual is not None and not (residual.shape == x.shape and residual.dtype == x.dtype
                                     and residual.device == x.device and residual.data_ptr() % 16 == 0
                                     and residual.is_contiguous(memory_format=torch.channels_last)):
        return None
    with torch.cuda.device(x.device):   # the launch goes to x's device and its current stream
        rc = _cabi.lib().tf_bias_act_f32(x.data_ptr(), bias.data_ptr(),
                                         0 if residual is None else residual.data_ptr(), x.numel(), C,
                                         1 if relu else 0, _stream(x.device))
    _cabi.check(rc, "tf_bias_act_f32")
    return x


def add_layernorm(x, res, norm):
    """LayerNorm(x + res) with `norm`'s affine parameters in one pass (res may be None).
    Returns a new tensor, or None when the fused kernel does not apply."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            and norm.elementwise_affine and len(norm.normalized_shape) == 1):
        return None
    C = norm.normalized_shape[0]
    if x.shape[-1] != C or C % 4 or C > 4096 or x.data_ptr() % 16:
        return None
    if not (_param_ok(norm.weight, x) and _param_ok(norm.bias, x)):
        return None
    if res is not None:
        if res.shape != x.shape or res.dtype != x.dtype or res.device != x.device:
            return None
        res = res.contiguous()
        if res.data_ptr() % 16:
            return None
    with torch.cuda.device(x.device):
        out = torch.empty_like(x)
        rc = _cabi.lib().tf_add_layernorm_f32(x.data_ptr(), 0 if res is None else res.data_ptr(),
                                              norm.weight.data_ptr(), norm.bias.data_ptr(),
                                              out.data_ptr(), x.numel() // C, C, float(norm.eps),
                                              _stream(x.device))
    _cabi.check(rc, "tf_add_layernorm_f32")
    return out


def residual_norm(x, res, norm, inference):
    """norm(x + res): fused on the GPU inference path, plain PyTorch otherwise."""
    if inference:
        y = add_layernorm(x, res, norm)
        if y is not None:
            return y
    return norm(x + res)


# ---------------------------------------------------------------------------------------------------------
# nn.Linear as a bf16 split product on the matrix cores (include/tf_fused.h: tf_linear_split_f32).  ON by default
# in the GPU inference path (TF_SPLIT_LINEAR=0 / set_split_linear(False): the fp32 library GEMMs): faster than the
# tuned hipBLASLt selections at the encoder shapes, 172.3 -> 173.2 frames/s end to end (4 sequences; 128.6 -> 131.3
# with one), and inside the parity bar at the BASELINE sizes (tests/test_full_size_gpu.py: boxes 1e-6, logits 6e-5,
# track ids exact).
_split_linear = os.environ.get("TF_SPLIT_LINEAR", "1") not in ("", "0")


def split_linear_enabled():
    return _split_linear


def set_split_linear(on):
    """Switch the split-product linears on or off (process-wide); returns the previous setting."""
    global _split_linear
    prev, _split_linear = _split_linear, bool(on)
    return prev


# The arithmetic of a split product (include/tf_fused.h, THE SPLIT PRODUCT; csrc/split_product.h):
#   6   bf16 pieces (hi, mid, lo) per operand, six terms: all 24 significand bits, dropped terms below 2^-24 of the product -- the
#       reference's fp32 arithmetic in another summation order.
#   16  fp16 pieces: (hi, lo) of the activation x (hi, lo, hi 2^-11) of the weight, THREE terms: 22 + 1 significand bits, the only
#       dropped product below 2^-22; weights scaled per output channel and the lo piece stored times 2^11 so that nothing falls
#       into fp16's subnormals.  Half the matrix work of 6 at fp32-class accuracy: THE DEFAULT (round 4).  MI355X, cfg 2: 396.9
#       frames/s against 295.6 with six terms (410.3 with three bf16 terms, 193.9 on the fp32 libraries); full-size logits 1.2e-5 /
#       boxes 5.4e-7 from the reference (six terms: 1.6e-5 / 5.4e-7); the reference's track ids on all 64 frames of the 64-frame
#       fixture in both runs (six terms: 26 frames, the fp32 libraries 59 / 26): profiles/r04_id_parity_64_with_fp16.txt.
# (3 = bf16 pieces (hi, mid), three terms, products good to 2^-16 -- the "fast mode" of rounds 2-4 -- was REMOVED in round 5: the
# fp16 product runs at the same speed, and the 64-frame reference-Tracker fixture keeps the reference's track ids for 59 frames
# under fp32 library GEMMs but only 14 under three bf16 terms, profiles/r04_id_parity_64.txt.)
def _parse_terms(v):
    v = str(v).strip().lower()
    if v in ("6", "16"):
        return int(v)
    if v in ("f16", "fp16", "half"):
        return 16
    if v == "3":
        raise ValueError("split terms 3 (the three-term bf16 fast mode) was removed in round 5: use 16 (fp16 pieces, the default: "
                         "same speed, fp32-class accuracy) or 6 (six bf16 terms)")
    raise ValueError("split terms: 6 or 16 (fp16 pieces)")


_split_terms = _parse_terms(os.environ.get("TF_SPLIT_TERMS", "16"))
_tls = threading.local()   # .terms: a routed layer's six-term product, for the CALLING thread only (see _ranged)


def split_terms():
    return _split_terms


def _terms():
    """The split product of the call being made: the calling thread's override (a layer routed to six terms) or the process-wide
    setting.  Every kernel call and every weight-image cache of this module reads it through here."""
    return getattr(_tls, "terms", None) or _split_terms


def set_split_terms(n):
    """6 (bf16 pieces, six terms) or 16 (fp16 pieces, three terms) per split product (process-wide; cached weight images are kept per
    setting); returns the previous value."""
    global _split_terms
    n = _parse_terms(n)
    prev, _split_terms = _split_terms, n
    return prev


# The same product with the weight packed once in matrix-core fragment order (tf_linear_packed_f32,
# csrc/linear_stream.hip): bit-identical results.  Used where it measured faster (profiles/r02_split_gemm_packed.txt):
# many rows and a wide output or a long K -- the FFN linears of the encoder (22 223 x 256 -> 1024: 76.2 -> 59.3 us,
# 1024 -> 256: 63.6 -> 50.5 us).  TF_LINEAR_PACKED=0 / set_packed_linear(False): the unpacked kernel everywhere.
_packed_linear = os.environ.get("TF_LINEAR_PACKED", "1") not in ("", "0")
_PACKED_MIN_ROWS = 4096


def _use_packed(M, K, N):
    """Shapes tf_linear_packed_f32 (the stream GEMM: weight fragments from L2, only the activations through LDS) is the faster
    kernel for.  Its column blocks (64 / 128 / 256 wide) must be mostly full.  Six terms: every many-row shape with K >= 256 -- at
    equal matrix work it beats the LDS-staged block kernel there (22 223 rows: 256 -> 256 23.9 vs 26.3 us, 256 -> 1024 67.5 vs
    84.7, 1024 -> 256 65.6 vs 88.0, profiles/r04_pmc_dense_six_terms.txt; the reducing 1 x 1 convolutions of ResNet-50 25.2 vs
    35.6 / 24.8 vs 28.9 / 35.9 vs 42.2 us) but not under a short K, where its prologue is most of the block (the expanding 1 x 1
    convolutions with their residual: K = 64: 34.8 vs 30.4 us, K = 128: 28.8 vs 26.0; profiles/r04_conv_per_layer_stream_vs_block.txt).
    Two stored weight pieces (the fp16 product): a wide output or a long K only (at K = 256, N <= 384 the many small blocks of the
    block kernel hide the memory latency better: 21.4 vs 22.8 us)."""
    if not _packed_linear or M <= _PACKED_MIN_ROWS or K % 64:
        return False
    tail = N % 256
    full = N <= 128 or tail == 0 or tail > 128
    if _terms() == 6:
        return full and K >= 256
    # two stored weight pieces (fp16 pieces; measured: profiles/r04_f16_harness.txt: 22 223 x 256 -> 256 18.4 vs 18.7 us, 16 700 x 512 -> 128 16.1 vs 21.6, 66 800 x 64 -> 256
    # 20.8 vs 19.3, 16 700 x 128 -> 512 17.3 vs 17.4)
    return (N >= 512 or K >= 512) and full


def set_packed_linear(on):
    """Switch the packed-weight kernel for the many-row linears on or off (process-wide); returns the previous setting."""
    global _packed_linear
    prev, _packed_linear = _packed_linear, bool(on)
    return prev


def _publish_barrier(device):
    """Cached weight images are built on the caller's current stream and then published on the tensor for EVERY stream
    (one tracker thread per sequence, each on its own HIP stream, share one model): wait for the building stream before
    publishing, so that a second thread's first GEMM cannot read a half-written image.  Once per weight."""
    if getattr(_tls, "one_stream", 0) and os.environ.get("TF_TRAIN_PUBLISH_BARRIER") != "1":
        return   # (a training step: the images are rebuilt every step for the stream that uses them next, see one_stream())
    if not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(device).synchronize()


class one_stream:
    """with fused.one_stream(): weight images built inside are used by the calling thread's current stream only -- no host
    synchronisation when they are published.  The previous-frame pass of a TRAINING step (no_grad, inference kernels: backbone.
    _fold_mode) rebuilds the images of every trainable convolution after every optimiser step; ~80 stream synchronisations per
    step otherwise."""

    def __enter__(self):
        _tls.one_stream = getattr(_tls, "one_stream", 0) + 1

    def __exit__(self, *exc):
        _tls.one_stream -= 1
        return False


def _packed_weight(weight, rows):
    """Fragment-order image (the bf16 pieces of the current split_terms()) of `weight` (or of its row block `rows`), built by
    tf_linear_pack_weight_f32 and cached on the tensor object with its version counter, like _split_weight below."""
    cache = getattr(weight, "_tf_packed", None)
    if cache is None or cache[0] != weight._version:
        cache = (weight._version, {})
        weight._tf_packed = cache
    terms = _terms()
    hit = cache[1].get((rows, terms))
    if hit is None:
        if torch.cuda.is_current_stream_capturing():
            return None   # never build a cached buffer inside a graph's memory pool: this call takes the unpacked kernel
        N, K = weight.shape
        a, b = (0, N) if rows is None else rows
        nbytes = _cabi.lib().tf_linear_packed_bytes(K, b - a, terms)
        if nbytes <= 0:
            return None
        w = weight.detach()
        with torch.cuda.device(weight.device):
            hit = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
            rc = _cabi.lib().tf_linear_pack_weight_f32(w.data_ptr() + a * K * 4, hit.data_ptr(), K, b - a, terms,
                                                       _stream(weight.device))
        _cabi.check(rc, "tf_linear_pack_weight_f32")
        _publish_barrier(weight.device)
        cache[1][(rows, terms)] = hit
    return hit


def _split_weight(weight):
    """The 16-bit pieces of an fp32 weight [N, K] for the kernels that take them as separate tensors (include/tf_fused.h, THE
    SPLIT PRODUCT) -> (p0, p1, p2, scale):
      split_terms() 6       bf16 hi, mid, lo; scale None
      split_terms() 16      fp16 wh = f16(w t_n), wl = f16(w t_n - wh), None, and scale[n] = 16 / t_n (fp32), t_n the power of two
                            that puts the largest |w| of output channel n into [2^13, 2^14)
    (round to nearest even at every step).  Cached ON THE TENSOR OBJECT together with its version counter
    (weights are constants in inference; an in-place update bumps the version).  Not keyed by data_ptr: a freed
    parameter's address is handed to the next model's parameters by the caching allocator, and a pointer-keyed cache
    then serves another tensor's pieces (seen as a golden failure when two test models were built one after the
    other).  Callers pass persistent tensors (module parameters, _CatProjection's concatenation), and use `rows=` of
    linear() for a row block instead of a temporary slice."""
    if _terms() == 16:
        hit = getattr(weight, "_tf_split_f16", None)
        if hit is None or hit[0] != weight._version:
            w = weight.detach()
            amax = w.abs().amax(dim=1)
            _, e = torch.frexp(amax)                                   # amax = m 2^e with m in [0.5, 1): floor(log2 amax) = e - 1
            # 2^(14 - e) from its exponent bits (torch.ldexp goes through pow(), which is not exact on every device)
            t = ((torch.clamp(14 - e, -100, 100).to(torch.int32) + 127) << 23).view(torch.float32)
            t = torch.where((amax > 0) & (amax < 3.0e38), t, torch.ones_like(t))
            ws = w * t[:, None]                                        # exact: powers of two
            hi = ws.to(torch.float16)
            lo = (ws - hi.float()).to(torch.float16)
            hit = (weight._version, hi.contiguous(), lo.contiguous(), None, (16.0 / t).contiguous())
            if w.is_cuda and torch.cuda.is_current_stream_capturing():
                return hit[1], hit[2], hit[3], hit[4]   # built inside a graph's memory pool: part of the graph, never a cached buffer
            if w.is_cuda:
                _publish_barrier(w.device)
            weight._tf_split_f16 = hit
        return hit[1], hit[2], hit[3], hit[4]
    hit = getattr(weight, "_tf_split", None)
    if hit is None or hit[0] != weight._version:
        w = weight.detach()
        hi = w.to(torch.bfloat16)
        r = w - hi.float()                      # exact in fp32
        mid = r.to(torch.bfloat16)
        lo = (r - mid.float()).to(torch.bfloat16).contiguous()
        hit = (weight._version, hi.contiguous(), mid.contiguous(), lo)
        if w.is_cuda and torch.cuda.is_current_stream_capturing():
            return hit[1], hit[2], hit[3], None   # (as above: not cached)
        if w.is_cuda:
            _publish_barrier(w.device)
        weight._tf_split = hit
    return hit[1], hit[2], hit[3], None


def _rows_of(pieces, rows):
    """The row block `rows` = (a, b) of _split_weight's pieces (views: a row block is contiguous)."""
    return tuple(None if p is None else p[rows[0]:rows[1]] for p in pieces)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def linear(x, weight, bias=None, relu=False, rows=None, residual=None):
    """act(x @ weight^T + bias) through tf_linear_split_f32 for fp32 GPU tensors with K % 32 == 0; returns None when
    it does not apply (switched off, other dtype / device / shape): the caller keeps its PyTorch formulation.
    rows = (a, b): use output features a..b of `weight` only (a row block of a packed projection such as
    nn.MultiheadAttention.in_proj_weight); `bias` is then the matching slice.
    residual: fp32 [rows of x, N], added before the activation (tf_linear_split_res_f32)."""
    if not (_split_linear and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and weight.dim() == 2 and weight.is_contiguous() and weight.device == x.device):
        return None
    N, K = weight.shape
    if rows is not None:
        if not (0 <= rows[0] < rows[1] <= N):
            return None
        N = rows[1] - rows[0]
    if x.shape[-1] != K or K % 32 or x.numel() == 0:
        return None
    if bias is not None and not (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N
                                 and bias.device == x.device):
        return None
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    if residual is not None and not (residual.dtype == torch.float32 and residual.is_contiguous() and residual.device == x.device
                                     and residual.numel() == x2.shape[0] * N):
        return None
    if _use_packed(x2.shape[0], K, N) and not (x2.data_ptr() & 15) and (x2.shape[0] + 256) * N * 4 < 0xC0000000:
        packed = _packed_weight(weight, rows)
        if packed is not None:
            with torch.cuda.device(x.device):
                y = torch.empty((x2.shape[0], N), dtype=torch.float32, device=x.device)
                rc = _cabi.lib().tf_linear_packed_f32(x2.data_ptr(), packed.data_ptr(),
                                                      0 if bias is None else bias.data_ptr(), _ptr(residual), y.data_ptr(),
                                                      x2.shape[0], K, N, 1 if relu else 0, _terms(), _stream(x.device))
            _cabi.check(rc, "tf_linear_packed_f32")
            return y.view(*x.shape[:-1], N)
    pieces = _split_weight(weight)
    if rows is not None:
        pieces = _rows_of(pieces, rows)
    hi, mid, lo, wsc = pieces
    if (x2.data_ptr() | hi.data_ptr() | mid.data_ptr() | _ptr(lo)) & 15:
        return None
    with torch.cuda.device(x.device):
        y = torch.empty((x2.shape[0], N), dtype=torch.float32, device=x.device)
        if residual is None:
            rc = _cabi.lib().tf_linear_split_f32(x2.data_ptr(), hi.data_ptr(), mid.data_ptr(), _ptr(lo), _ptr(wsc),
                                                 0 if bias is None else bias.data_ptr(), y.data_ptr(), x2.shape[0], K, N,
                                                 1 if relu else 0, _stream(x.device))
        else:
            rc = _cabi.lib().tf_linear_split_res_f32(x2.data_ptr(), hi.data_ptr(), mid.data_ptr(), _ptr(lo), _ptr(wsc),
                                                     0 if bias is None else bias.data_ptr(), residual.data_ptr(),
                                                     y.data_ptr(), x2.shape[0], K, N, 1 if relu else 0, _stream(x.device))
    _cabi.check(rc, "tf_linear_split_f32")
    return y.view(*x.shape[:-1], N)


# DEFAULT since round 3 (TF_FFN_FUSED=0 / set_ffn_fused(False) switches it off): linear1 -> ReLU -> linear2 -> + residual -> LayerNorm of a transformer
# layer in ONE launch (tf_ffn_fused_f32, csrc/ffn_fused.hip): the d_ffn-wide intermediate (91 MB per encoder layer at
# 800 x 1333) never leaves the CU.  Same split products as linear(), bit-identical before the LayerNorm.  MI355X, 22 223 rows, d_ffn 1024: 72.3 us against
# 116.0 us for linear1 + ReLU, linear2, residual + LayerNorm (profiles/r03_optin_ffn_fused.txt).
_ffn_fused = os.environ.get("TF_FFN_FUSED", "1") != "0"
_FFN_FUSED_MIN_ROWS = int(os.environ.get("TF_FFN_FUSED_MIN_ROWS", "4096"))   # below: too few 96-row blocks to fill the CUs


def ffn_fused_enabled():
    return _ffn_fused


def set_ffn_fused(on):
    """Switch the one-launch feed-forward block on or off (process-wide); returns the previous setting."""
    global _ffn_fused
    prev, _ffn_fused = _ffn_fused, bool(on)
    return prev


def ffn(x, linear1, linear2, norm=None, residual=None):
    """[norm](residual + linear2(relu(linear1(x)))) through tf_ffn_fused_f32 (reference: deformable_transformer.py:282-297
    forward_ffn + norm2).  x [..., 256 | 288] fp32 on the GPU; linear1 / linear2: nn.Linear; norm: nn.LayerNorm or None;
    residual: like x or None.  Returns None when the kernel does not apply (the caller keeps the separate kernels)."""
    if not (_ffn_fused and _split_linear and x.is_cuda and x.dtype == torch.float32):
        return None
    w1, w2 = linear1.weight, linear2.weight
    F_, D = w1.shape
    if D not in (256, 288) or x.shape[-1] != D or tuple(w2.shape) != (D, F_) or F_ % 16 or F_ < 128 or x.numel() == 0:
        return None
    if not all(w.dtype == torch.float32 and w.is_contiguous() and w.device == x.device for w in (w1, w2)):
        return None
    x2 = x.reshape(-1, D)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    if M < _FFN_FUSED_MIN_ROWS or (M + 128) * D * 4 > 0xFFFFFFFF:
        return None
    vecs = [linear1.bias, linear2.bias]
    if norm is not None:
        if not (norm.elementwise_affine and tuple(norm.normalized_shape) == (D,)):
            return None
        vecs += [norm.weight, norm.bias]
    if not all(v is None or _param_ok(v, x) for v in vecs):
        return None
    if residual is not None:
        if residual.dtype != torch.float32 or residual.device != x.device or residual.numel() != x2.numel():
            return None
        residual = residual.reshape(-1, D)
        if not residual.is_contiguous():
            residual = residual.contiguous()
    if (x2.data_ptr() | (0 if residual is None else residual.data_ptr())) & 15:
        return None
    p1, p2 = _packed_weight(w1, None), _packed_weight(w2, None)
    if p1 is None or p2 is None:
        return None
    ptr = lambda t: 0 if t is None else t.data_ptr()
    with torch.cuda.device(x.device):
        y = torch.empty((M, D), dtype=torch.float32, device=x.device)
        rc = _cabi.lib().tf_ffn_fused_f32(x2.data_ptr(), p1.data_ptr(), ptr(linear1.bias), p2.data_ptr(), ptr(linear2.bias),
                                          ptr(residual), 0 if norm is None else norm.weight.data_ptr(),
                                          0 if norm is None else norm.bias.data_ptr(), 0.0 if norm is None else float(norm.eps),
                                          y.data_ptr(), M, D, F_, _terms(), _stream(x.device))
    _cabi.check(rc, "tf_ffn_fused_f32")
    return y.view(x.shape)


# DEFAULT since round 3 (TF_LINLN_FUSED=0 / set_linear_ln_fused(False) switches it off): a 256 -> 256 linear, the layer's residual add and its LayerNorm in
# ONE launch (tf_linear_res_ln_f32, csrc/ffn_fused.hip) -- the attention's output projection + norm1.  MI355X: 19.4-22.1 us
# against 34.5 us at 22 223 rows, 6.9 against 17.1 us at 400 rows (profiles/r03_optin_ffn_fused.txt).
_linln_fused = os.environ.get("TF_LINLN_FUSED", "1") != "0"
_LINLN_MIN_ROWS = int(os.environ.get("TF_LINLN_MIN_ROWS", "256"))


def linear_ln_fused_enabled():
    return _linln_fused


def set_linear_ln_fused(on):
    """Switch the one-launch projection + residual + LayerNorm on or off (process-wide); returns the previous setting."""
    global _linln_fused
    prev, _linln_fused = _linln_fused, bool(on)
    return prev


def linear_residual_norm(x, linear, residual, norm):
    """norm(residual + linear(x)) through tf_linear_res_ln_f32 for a 256 -> 256 or 288 -> 288 nn.Linear (reference:
    ms_deform_attn.py:87 output_proj + deformable_transformer.py:285-292).  Returns None when the kernel does not apply."""
    if not (_linln_fused and _split_linear and x.is_cuda and x.dtype == torch.float32):
        return None
    w = linear.weight
    D = w.shape[0]
    if D not in (256, 288) or tuple(w.shape) != (D, D) or x.shape[-1] != D or x.numel() == 0:
        return None
    if not (w.dtype == torch.float32 and w.is_contiguous() and w.device == x.device
            and norm.elementwise_affine and tuple(norm.normalized_shape) == (D,)):
        return None
    x2 = x.reshape(-1, D)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    if M < _LINLN_MIN_ROWS or (M + 128) * D * 4 > 0xFFFFFFFF:
        return None
    if not all(v is None or _param_ok(v, x) for v in (linear.bias, norm.weight, norm.bias)):
        return None
    if residual.dtype != torch.float32 or residual.device != x.device or residual.numel() != x2.numel():
        return None
    residual = residual.reshape(-1, D)
    if not residual.is_contiguous():
        residual = residual.contiguous()
    if (x2.data_ptr() | residual.data_ptr()) & 15:
        return None
    packed = _packed_weight(w, None)
    if packed is None:
        return None
    with torch.cuda.device(x.device):
        y = torch.empty((M, D), dtype=torch.float32, device=x.device)
        rc = _cabi.lib().tf_linear_res_ln_f32(x2.data_ptr(), packed.data_ptr(), 0 if linear.bias is None else linear.bias.data_ptr(),
                                              residual.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps),
                                              y.data_ptr(), M, D, D, _terms(), _stream(x.device))
    _cabi.check(rc, "tf_linear_res_ln_f32")
    return y.view(x.shape)


# DEFAULT since round 3 (TF_STEM_POOL_FUSED=0 / set_stem_pool_fused(False) switches it off; stem 147.4 -> 66.4 us with the
# convolution below, profiles/r03_optin_conv_per_layer.txt): FrozenBN shift + ReLU + MaxPool2d(3, 2, 1) after the stem
# convolution in one pass (tf_bias_relu_maxpool_f32) instead of bias_act_ + F.max_pool2d; bit-identical.
_stem_pool_fused = os.environ.get("TF_STEM_POOL_FUSED", "1") != "0"


def stem_pool_fused_enabled():
    return _stem_pool_fused


def set_stem_pool_fused(on):
    global _stem_pool_fused
    prev, _stem_pool_fused = _stem_pool_fused, bool(on)
    return prev


# DEFAULT since round 3 (TF_STEM_CONV_SPLIT=0 / set_stem_conv_split(False) switches it off): the 7 x 7 stem convolution as a split product on the matrix cores
# (tf_stem_conv7x7_f32, csrc/stem_conv.hip) instead of the library convolution.
_stem_conv_split = os.environ.get("TF_STEM_CONV_SPLIT", "1") != "0"


def stem_conv_split_enabled():
    return _stem_conv_split


def set_stem_conv_split(on):
    global _stem_conv_split
    prev, _stem_conv_split = _stem_conv_split, bool(on)
    return prev


def _stem_packed(weight):
    """Packed [64, 176] image (k = (c * 7 + ky) * 8 + kx, zero padded) of a [64, 3, 7, 7] weight, cached on the tensor."""
    hit = getattr(weight, "_tf_stem_packed", None)
    terms = _terms()
    if hit is None or hit[0] != (weight._version, terms):
        if torch.cuda.is_current_stream_capturing():
            return None
        w = weight.detach()
        w2 = torch.zeros((64, 176), dtype=torch.float32, device=w.device)
        w2[:, :168] = F.pad(w, (0, 1)).reshape(64, 168)
        nbytes = _cabi.lib().tf_linear_packed_bytes(176, 64, terms)
        with torch.cuda.device(w.device):
            packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
            rc = _cabi.lib().tf_linear_pack_weight_f32(w2.data_ptr(), packed.data_ptr(), 176, 64, terms, _stream(w.device))
        _cabi.check(rc, "tf_linear_pack_weight_f32")
        _publish_barrier(w.device)
        hit = ((weight._version, terms), packed)
        weight._tf_stem_packed = hit
    return hit[1]


def stem_conv(x, weight, bias=None, relu=False):
    """conv2d(x, weight, stride 2, padding 3) (+ bias, ReLU) for x [N, 3, H, W] fp32 on the GPU and a [64, 3, 7, 7] weight
    (reference: torchvision resnet50.conv1 under models/backbone.py:93-104) -> channels_last [N, 64, Ho, Wo], or None."""
    if not (_stem_conv_split and _split_linear and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3
            and tuple(weight.shape) == (64, 3, 7, 7) and weight.dtype == torch.float32 and weight.device == x.device
            and x.numel() > 0):
        return None
    if bias is not None and not (_param_ok(bias, x) and bias.numel() == 64):
        return None
    x = x if x.is_contiguous() else x.contiguous()
    n, _, h, w = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if n > 65535 or n * ho * wo * 256 >= 0xC0000000 or n * 3 * h * w >= 2 ** 31:
        return None
    packed = _stem_packed(weight)
    if packed is None:
        return None
    with torch.cuda.device(x.device):
        y = torch.empty((n, 64, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        rc = _cabi.lib().tf_stem_conv7x7_f32(x.data_ptr(), packed.data_ptr(), 0 if bias is None else bias.data_ptr(), y.data_ptr(),
                                             n, h, w, 1 if relu else 0, _terms(), _stream(x.device))
    _cabi.check(rc, "tf_stem_conv7x7_f32")
    return y


def bias_relu_maxpool(x, bias):
    """maxpool3x3/s2/p1(relu(x + bias[c])) of a channels_last [N, C, H, W] fp32 GPU tensor -> channels_last
    [N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1], or None when the kernel does not apply."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and x.numel() > 0 and x.shape[1] % 4 == 0 and _param_ok(bias, x) and bias.numel() == x.shape[1]
            and x.data_ptr() % 16 == 0):
        return None
    n, c, h, w = x.shape
    with torch.cuda.device(x.device):
        out = torch.empty((n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=torch.float32, device=x.device,
                          memory_format=torch.channels_last)
        rc = _cabi.lib().tf_bias_relu_maxpool_f32(x.data_ptr(), bias.data_ptr(), out.data_ptr(), n, h, w, c, _stream(x.device))
    _cabi.check(rc, "tf_bias_relu_maxpool_f32")
    return out


# DEFAULT since round 3 (TF_HEADS_SPLIT=0 / set_heads_split(False) switches it off; +2.9 % frames/s on its own,
# profiles/r03_optin_single_routes.txt): the small head GEMMs (box-regression MLPs, class heads: 400 rows) through
# the split-product kernels as well instead of hipBLASLt -- 24 launches per frame; the class logits then carry the split
# product's fp32-class error like everything else (parity and track ids: tests/test_full_size_gpu.py).
_heads_split = os.environ.get("TF_HEADS_SPLIT", "1") != "0"


def heads_split_enabled():
    return _heads_split


def set_heads_split(on):
    global _heads_split
    prev, _heads_split = _heads_split, bool(on)
    return prev


def head_linear(module, x, relu=False):
    """act(module(x)) for a head's nn.Linear: the split product when the opt-in is on (inference on the GPU), else the module."""
    if _heads_split and not module.training and not torch.is_grad_enabled():
        y = linear(x, module.weight, module.bias, relu=relu)
        if y is not None:
            return y
    y = module(x)
    return F.relu(y) if relu else y


# DEFAULT since round 3 (TF_POS_ADD_FUSED=0 / set_pos_add_fused(False) switches it off): `with_pos_embed(x, pos)` in front of a projection is done inside the
# GEMM while the activation tile is staged (tf_linear_split_add_f32) instead of as its own pass over the tokens; bit-identical.
_pos_add_fused = os.environ.get("TF_POS_ADD_FUSED", "1") != "0"


def pos_add_fused_enabled():
    return _pos_add_fused


def set_pos_add_fused(on):
    global _pos_add_fused
    prev, _pos_add_fused = _pos_add_fused, bool(on)
    return prev


def linear_add(x, x2, weight, bias=None, rows=None):
    """(x + x2) @ weight^T + bias through tf_linear_split_add_f32 (same conditions and `rows` meaning as linear());
    returns None when it does not apply."""
    if not (_pos_add_fused and _split_linear and x.is_cuda and x.dtype == torch.float32 and x2.dtype == torch.float32
            and x2.shape == x.shape and x2.device == x.device and weight.dtype == torch.float32 and weight.dim() == 2
            and weight.is_contiguous() and weight.device == x.device):
        return None
    N, K = weight.shape
    if rows is not None:
        if not (0 <= rows[0] < rows[1] <= N):
            return None
        N = rows[1] - rows[0]
    if x.shape[-1] != K or K % 32 or x.numel() == 0:
        return None
    if bias is not None and not (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N
                                 and bias.device == x.device):
        return None
    a, b = x.reshape(-1, K), x2.reshape(-1, K)
    a = a if a.is_contiguous() else a.contiguous()
    b = b if b.is_contiguous() else b.contiguous()
    pieces = _split_weight(weight)
    if rows is not None:
        pieces = _rows_of(pieces, rows)
    hi, mid, lo, wsc = pieces
    if (a.data_ptr() | b.data_ptr() | hi.data_ptr() | mid.data_ptr() | _ptr(lo)) & 15:
        return None
    with torch.cuda.device(x.device):
        y = torch.empty((a.shape[0], N), dtype=torch.float32, device=x.device)
        rc = _cabi.lib().tf_linear_split_add_f32(a.data_ptr(), b.data_ptr(), hi.data_ptr(), mid.data_ptr(), _ptr(lo), _ptr(wsc),
                                                 0 if bias is None else bias.data_ptr(), y.data_ptr(), a.shape[0], K, N,
                                                 _stream(x.device))
    _cabi.check(rc, "tf_linear_split_add_f32")
    return y.view(*x.shape[:-1], N)


def conv3x3(x, w_taps, bias, relu, stride):
    """3 x 3 convolution (padding 1) -- or, with a [Cout, Cin] weight, a strided 1 x 1 convolution without padding -- of a
    channels_last fp32 GPU activation through tf_conv3x3_split_f32 / tf_conv1x1_strided_split_f32.
    x [N, Cin, H, W] (channels_last), w_taps [Cout, 9 * Cin] (the [Cout, 3, 3, Cin] storage of a channels_last weight, a
    persistent tensor: its bf16 pieces are cached on it), bias [Cout] or None.  Returns [N, Cout, Hout, Wout]
    (channels_last) or None when the kernel does not apply."""
    if not (_split_linear and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and w_taps.dtype == torch.float32
            and w_taps.dim() == 2 and w_taps.is_contiguous() and w_taps.device == x.device):
        return None
    n, cin, h, w = x.shape
    cout = w_taps.shape[0]
    ks = 3 if w_taps.shape[1] == 9 * cin else 1   # [Cout, Cin]: a strided 1 x 1 projection (no padding)
    if w_taps.shape[1] != ks * ks * cin or cin % 32 or stride not in (1, 2) or x.numel() == 0:
        return None
    if bias is not None and not (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == cout
                                 and bias.device == x.device):
        return None
    pad = 1 if ks == 3 else 0
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    # (the stream form takes Cin % 64; its halo form -- stride-1 3 x 3 -- Cin % 32: the mask head's 288- and 32-channel layers)
    cin_ok = cin % 64 == 0 or (ks == 3 and stride == 1 and _conv_halo and cin % 32 == 0)
    stream_route = (_conv_stream and _conv_stream_wins(n * ho * wo, cout) and cin_ok and not (x.data_ptr() & 15)
                    and n * h * w * cin * 4 < 0xC0000000 and (n * ho * wo + 256) * cout * 4 < 0xC0000000)
    halo = ks == 3 and stride == 1 and _conv_halo and stream_route   # (the halo form of the stream route: its own split policy)
    ksplit = _conv_ksplit(n * ho * wo, ks * ks * cin, cout, _HALO_KSPLIT_POLICY if halo else None)
    if ks == 1 and (ksplit < _CONV1X1_MIN_PIECES or not _conv1x1_splitk):
        ksplit = 1
    if stream_route:
        packed = _packed_weight(w_taps, None)
        if packed is not None:
            with torch.cuda.device(x.device):
                y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
                ws = torch.empty((ksplit, n * ho * wo * cout), dtype=torch.float32, device=x.device) if ksplit > 1 else None
                rc = _cabi.lib().tf_conv_packed_f32(x.data_ptr(), packed.data_ptr(), _ptr(bias), 0, y.data_ptr(), _ptr(ws), ksplit,
                                                    n, h, w, cin, cout, ks, stride, 1 if relu else 0, _terms(), _stream(x.device))
            _cabi.check(rc, "tf_conv_packed_f32")
            return y.permute(0, 3, 1, 2)   # NCHW shape over NHWC storage = channels_last
    # the block kernels address input, output and weight pieces through buffer resources: every byte offset below 3 GiB
    if (n * h * w * cin * 4 >= 0xC0000000 or (n * ho * wo + 256) * cout * 4 >= 0xC0000000 or cout * ks * ks * cin * 2 >= 0xC0000000):
        return None
    hi, mid, lo, wsc = _split_weight(w_taps)
    if (x.data_ptr() | hi.data_ptr() | mid.data_ptr() | _ptr(lo)) & 15:
        return None
    with torch.cuda.device(x.device):
        y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
        if ksplit > 1:   # few output pixels under a long K: split the K loop over workgroups (deterministic second pass)
            ws = torch.empty((ksplit, n * ho * wo * cout), dtype=torch.float32, device=x.device)
            fn = _cabi.lib().tf_conv3x3_splitk_f32 if ks == 3 else _cabi.lib().tf_conv1x1_splitk_f32
            rc = fn(x.data_ptr(), hi.data_ptr(), mid.data_ptr(), _ptr(lo), _ptr(wsc), 0 if bias is None else bias.data_ptr(), y.data_ptr(),
                    ws.data_ptr(), ksplit, n, h, w, cin, cout, stride, 1 if relu else 0, _stream(x.device))
        else:
            fn = _cabi.lib().tf_conv3x3_split_f32 if ks == 3 else _cabi.lib().tf_conv1x1_strided_split_f32
            rc = fn(x.data_ptr(), hi.data_ptr(), mid.data_ptr(), _ptr(lo), _ptr(wsc), 0 if bias is None else bias.data_ptr(), y.data_ptr(),
                    n, h, w, cin, cout, stride, 1 if relu else 0, _stream(x.device))
    _cabi.check(rc, "tf_conv3x3_split_f32")
    return y.permute(0, 3, 1, 2)   # NCHW shape over NHWC storage = channels_last


_conv_splitk = os.environ.get("TF_CONV_SPLITK", "1") not in ("", "0")

# Round 4 (TF_CONV_STREAM=0 / set_conv_stream(False): the LDS-staged block kernel of linear_split.hip): the convolutions through
# the stream GEMM (tf_conv_packed_f32, csrc/linear_stream.hip) -- weight fragments straight from L2, only the shifted input
# pixels through LDS.  Same products in the same order: bit-identical without split-K.
_conv_stream = os.environ.get("TF_CONV_STREAM", "1") not in ("", "0")
_CONV_STREAM_ALL = os.environ.get("TF_CONV_STREAM") == "all"   # A/B aid / tests: every shape through the stream form


def _conv_stream_wins(m, cout):
    """Where the stream form of the 3 x 3 / strided convolutions measured faster than the LDS-staged block kernel on MI355X
    (profiles/r04_conv_per_layer_stream_vs_block.txt, 800 x 1333 frame): many output pixels under at least 128 output channels
    -- layer2's 3 x 3 convolutions (six terms 47.5 vs 51.3 us, three terms 30.6 vs 34.9), its strided first one (45.9 vs 51.2 /
    31.4 vs 36.4).  Not the 64-channel layer1 (two column tiles only: 47.7 vs 43.6 us) and not the few-pixel layers whose K loop
    is cut into pieces (layer3 / layer4: 49.0 vs 46.9, 48.8 vs 48.6) -- there the two forms are within noise or the block kernel
    wins, and it stays."""
    # fp16 pieces: every shape (per frame 1321 us with the stream form everywhere against 1362 with the block kernels,
    # profiles/r04_f16_conv_per_layer.txt)
    return _CONV_STREAM_ALL or _terms() == 16 or (cout >= 128 and m >= 8192)


def set_conv_stream(on):
    """True: where it measured faster (the default); "all": every shape (tests, A/B); False: never.  Returns the previous setting."""
    global _conv_stream, _CONV_STREAM_ALL
    prev = "all" if (_conv_stream and _CONV_STREAM_ALL) else _conv_stream
    _conv_stream, _CONV_STREAM_ALL = bool(on), on == "all"
    return prev



# Split-K policy of the 3 x 3 convolutions: (workgroups aimed at, launches with at least this many blocks are left alone,
# fewest K-slices a piece may have, fewest K-slices a layer must have).  Swept on the MI355X at the 800 x 1333 frame
# (profiles/r03_conv3_ksplit.txt; us of backbone convolutions per frame): 384,160,8,64 (rounds 2-3) 1473, 768,300,8,32 1396,
# 1024,300,6,32 1409, 1024,600,4,16 1418, 2048,600,4,16 1493.
_KSPLIT_POLICY = tuple(int(v) for v in os.environ.get("TF_CONV_KSPLIT_POLICY", "768,300,8,32").split(","))


def set_conv_ksplit_policy(target_blocks, leave_alone_blocks, min_slices_per_piece, min_slices):
    """-> the previous policy tuple."""
    global _KSPLIT_POLICY
    prev, _KSPLIT_POLICY = _KSPLIT_POLICY, (int(target_blocks), int(leave_alone_blocks), int(min_slices_per_piece), int(min_slices))
    return prev


def conv1x1_wants_split_k(m, cin, cout):
    """True when a stride-1 1 x 1 convolution over m pixels is better off in the convolution kernel with its K loop cut
    (tf_conv1x1_splitk_f32) than in the plain split GEMM: few workgroups, each walking a long K.  Default since round 3
    (TF_CONV1X1_SPLITK=0 / set_conv1x1_splitk(False) switches it off), measured at the 800 x 1333 frame
    (profiles/r03_conv1x1_splitk.txt): layer3.1-5.conv1 (1024 -> 256 at 50 x 84, 4 pieces) 30.2 -> 21.8 us, layer4.1-2.conv1
    (2048 -> 512 at 25 x 42, 8 pieces) 40.1 -> 23.9 us; two pieces did not pay (layer4.0.conv1 31.1 -> 32.5 us, the strided
    projection of layer4 33.6 -> 34.0 us): a 1 x 1 convolution is only cut into three pieces or more."""
    return _conv1x1_splitk and _conv_ksplit(m, cin, cout) >= _CONV1X1_MIN_PIECES


_conv1x1_splitk = os.environ.get("TF_CONV1X1_SPLITK", "1") not in ("", "0")
_CONV1X1_MIN_PIECES = 3


def set_conv1x1_splitk(on):
    global _conv1x1_splitk
    prev, _conv1x1_splitk = _conv1x1_splitk, bool(on)
    return prev


# The halo form of the stride-1 3 x 3 convolutions (csrc/linear_stream.hip conv3x3_halo_kernel, round 6; TF_CONV_HALO=0 /
# set_conv_halo(False): the stream form): a block stages the halo of its patch once per channel slice and runs all nine taps from it.
# Its blocks carry 18 x the matrix work per barrier of the stream form's, so it wants fewer, longer pieces: swept on the MI355X
# (profiles/r06_conv3_halo.txt).
_conv_halo = os.environ.get("TF_CONV_HALO", "1") not in ("", "0")
_HALO_KSPLIT_POLICY = tuple(int(v) for v in os.environ.get("TF_CONV_HALO_KSPLIT_POLICY", "512,300,9,32").split(","))


def set_conv_halo(on):
    """Switch the halo form of the stride-1 3 x 3 convolutions on or off (process-wide, also inside the library); returns the previous setting."""
    global _conv_halo
    prev, _conv_halo = _conv_halo, bool(on)
    _cabi.lib().tf_msda_set_option(b"conv_halo", 1 if on else 0)
    return prev


def _conv_ksplit(m, k, cout, policy=None):
    """Pieces the K loop (k = 9 Cin or Cin) of a split-product convolution is cut into: 1 unless the launch would leave most of the chip
    with one workgroup or none while every one of them walks a long K (at 800 x 1333: layer2's stride-1 layers 261
    workgroups of 36 K-slices -> 2 pieces, layer3 132 of 72 -> 5, layer4 68 of 144 -> 11, the extra pyramid level 10 of
    576 -> 64)."""
    if not _conv_splitk or cout % 4:
        return 1
    target, alone, per_piece, min_slices = policy or _KSPLIT_POLICY
    blocks = -(-m // 64) * -(-cout // (128 if cout >= 128 else 64))
    slices = k // 32
    if blocks >= alone or slices < min_slices:
        return 1
    return max(1, min(64, target // blocks, slices // per_piece))


# DEFAULT since round 3 (TF_INPUT_PROJ_FUSED=0 / set_input_proj_fused(False) switches it off; +4.3 % frames/s on its own,
# profiles/r03_optin_single_routes.txt): the reference's `input_proj`
# levels with a 1 x 1 convolution (models/deformable_detr.py:73-90: Conv2d(C, hidden, 1) -> GroupNorm(32, hidden)) as the
# split-product GEMM over the pixels + the library's own channels-innermost GroupNorm (tf_groupnorm_nhwc_f32), instead
# of a CK convolution, a layout copy and three ATen kernels per level.
_input_proj_fused = os.environ.get("TF_INPUT_PROJ_FUSED", "1") != "0"


def set_input_proj_fused(on):
    global _input_proj_fused
    prev, _input_proj_fused = _input_proj_fused, bool(on)
    return prev


def groupnorm_nhwc(x2, n_img, gn, relu=False):
    """GroupNorm [+ ReLU] of x2 [n_img * HW, C] (channels innermost) with nn.GroupNorm `gn`'s parameters; returns a new tensor of
    the same shape or None when the kernel does not apply."""
    if not (x2.is_cuda and x2.dtype == torch.float32 and x2.dim() == 2 and x2.is_contiguous()
            and gn.weight is not None and gn.bias is not None and gn.weight.device == x2.device):
        return None
    rows, c = x2.shape
    if rows % n_img or c != gn.num_channels or c % 4 or c > 1024 or gn.num_groups > 256 or n_img > 65535 or (x2.data_ptr() & 15):
        return None
    hw = rows // n_img
    with torch.cuda.device(x2.device):
        out = torch.empty_like(x2)
        ws = torch.empty(2 * n_img * gn.num_groups, dtype=torch.float64, device=x2.device)
        fn = _cabi.lib().tf_groupnorm_relu_nhwc_f32 if relu else _cabi.lib().tf_groupnorm_nhwc_f32
        rc = fn(x2.data_ptr(), gn.weight.data_ptr(), gn.bias.data_ptr(), out.data_ptr(), ws.data_ptr(), n_img, hw, c, gn.num_groups,
                float(gn.eps), hw * c, hw * c, _stream(x2.device))
    _cabi.check(rc, "tf_groupnorm_nhwc_f32")
    return out


def mask_label_map(logits, order, pad_hw, img_hw, out_hw, threshold=0.5):
    """The tracker's mask post-processing in one launch (tf_mask_label_map_f32): logits [n, h, w] fp32 on the GPU (mask-head outputs),
    order: per track the row of `logits` that is its mask (-1: none) -> int16 [out_h, out_w], the owning track per pixel or -1.
    None when not applicable."""
    if not (_postprocess_fused and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 3 and logits.is_contiguous()
            and 0 < len(order) <= 32767 and logits.numel() > 0):
        return None
    n, h, w = logits.shape
    if any(r >= n for r in order) or img_hw[0] > pad_hw[0] or img_hw[1] > pad_hw[1]:
        return None
    with torch.cuda.device(logits.device):
        order_dev = torch.tensor(list(order), dtype=torch.int32).to(logits.device, non_blocking=True)
        label = torch.empty((int(out_hw[0]), int(out_hw[1])), dtype=torch.int16, device=logits.device)
        rc = _cabi.lib().tf_mask_label_map_f32(logits.data_ptr(), order_dev.data_ptr(), label.data_ptr(), len(order), h, w, int(pad_hw[0]),
                                               int(pad_hw[1]), int(img_hw[0]), int(img_hw[1]), int(out_hw[0]), int(out_hw[1]), float(threshold),
                                               _stream(logits.device))
    _cabi.check(rc, "tf_mask_label_map_f32")
    return label


def upsample_add(low, fpn, q_per_image):
    """The mask head's FPN merge (detr_segmentation._merge) in one pass: F.interpolate(low, size=fpn.shape[-2:], mode="nearest") +
    fpn broadcast over the queries of its image.  low [N, C, h, w], fpn [N / q_per_image, C, H, W], both channels_last -> [N, C, H, W]
    channels_last, bit-identical to the two ATen passes; None when not applicable."""
    if not (low.is_cuda and low.dtype == torch.float32 and fpn.dtype == torch.float32 and low.dim() == 4 and fpn.dim() == 4
            and fpn.device == low.device and low.shape[1] == fpn.shape[1] and low.shape[1] % 4 == 0 and q_per_image > 0
            and low.shape[0] == fpn.shape[0] * q_per_image and low.is_contiguous(memory_format=torch.channels_last)
            and fpn.is_contiguous(memory_format=torch.channels_last) and not ((low.data_ptr() | fpn.data_ptr()) & 15)):
        return None
    n, c, h, w = low.shape
    H, W = fpn.shape[-2:]
    if n * H * W * c >= 1 << 33:
        return None
    with torch.cuda.device(low.device):
        out = torch.empty((n, H, W, c), dtype=torch.float32, device=low.device)
        rc = _cabi.lib().tf_upsample_add_nhwc_f32(low.data_ptr(), fpn.data_ptr(), out.data_ptr(), n, q_per_image, h, w, H, W, c, _stream(low.device))
    _cabi.check(rc, "tf_upsample_add_nhwc_f32")
    return out.permute(0, 3, 1, 2)   # NCHW shape over NHWC storage = channels_last


def groupnorm_stats(x, gn):
    """The statistics pass of GroupNorm alone for a channels_last x [N, C, H, W]: 2 * N * groups doubles (sum | sum of squares per image
    and group) for a consumer that normalises in its own fetch (conv3x3_merged).  None when not applicable."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and gn.num_channels == x.shape[1] and x.shape[1] % 4 == 0 and x.shape[1] <= 1024 and gn.num_groups <= 256
            and x.shape[0] <= 65535 and not (x.data_ptr() & 15)):
        return None
    n, c, h, w = x.shape
    with torch.cuda.device(x.device):
        ws = torch.empty(2 * n * gn.num_groups, dtype=torch.float64, device=x.device)
        rc = _cabi.lib().tf_groupnorm_stats_nhwc_f32(x.data_ptr(), ws.data_ptr(), n, h * w, c, gn.num_groups, h * w * c, _stream(x.device))
    _cabi.check(rc, "tf_groupnorm_stats_nhwc_f32")
    return ws


def conv3x3_merged(low, fpn, q_per_image, w_taps, bias, gn=None, ws=None):
    """3 x 3 / padding 1 convolution over  act(low) up-sampled (nearest) to fpn's size + fpn broadcast over the q_per_image queries of
    its image  -- the mask head's FPN merge -- computed in the convolution's fetch (tf_conv3x3_merge_packed_f32): the merged tensor is
    never written.  act = relu(gn(low)) from `ws` (groupnorm_stats of low) when gn is given, else identity.  low [N, Cin, h, w], fpn
    [N / q_per_image, Cin, H, W] channels_last, w_taps [Cout, 9 * Cin] tap-major (a persistent tensor).  -> [N, Cout, H, W] channels_last,
    or None when not applicable."""
    if not (_split_linear and _conv_halo and low.is_cuda and low.dtype == torch.float32 and fpn.dtype == torch.float32 and low.dim() == 4
            and fpn.dim() == 4 and fpn.device == low.device and low.shape[1] == fpn.shape[1] and q_per_image > 0
            and low.shape[0] == fpn.shape[0] * q_per_image and low.is_contiguous(memory_format=torch.channels_last)
            and fpn.is_contiguous(memory_format=torch.channels_last) and w_taps.dtype == torch.float32 and w_taps.dim() == 2
            and w_taps.is_contiguous() and w_taps.device == low.device and not ((low.data_ptr() | fpn.data_ptr()) & 15)):
        return None
    n, cin, lh, lw = low.shape
    H, W = fpn.shape[-2:]
    cout = w_taps.shape[0]
    if w_taps.shape[1] != 9 * cin or cin % 32 or cin > 320 or (gn is not None and (ws is None or gn.num_channels != cin or gn.weight is None)):
        return None
    if bias is not None and not (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == cout and bias.device == low.device):
        return None
    if (n * H * W + 256) * cout * 4 >= 0xC0000000 or n * lh * lw * cin * 4 >= 0xC0000000 or fpn.numel() * 4 >= 0xC0000000:
        return None
    packed = _packed_weight(w_taps, None)
    if packed is None:
        return None
    with torch.cuda.device(low.device):
        y = torch.empty((n, H, W, cout), dtype=torch.float32, device=low.device)
        rc = _cabi.lib().tf_conv3x3_merge_packed_f32(
            low.data_ptr(), fpn.data_ptr(), 0 if gn is None else ws.data_ptr(), 0 if gn is None else gn.weight.data_ptr(),
            0 if gn is None else gn.bias.data_ptr(), 1 if gn is None else gn.num_groups, 0.0 if gn is None else float(gn.eps), packed.data_ptr(),
            _ptr(bias), y.data_ptr(), n, q_per_image, lh, lw, H, W, cin, cout, 0, _terms(), _stream(low.device))
    _cabi.check(rc, "tf_conv3x3_merge_packed_f32")
    return y.permute(0, 3, 1, 2)   # NCHW shape over NHWC storage = channels_last


def groupnorm_relu_conv3x3_c1(x, gn, conv):
    """conv(relu(gn(x))) for a 3 x 3 / padding 1 convolution to ONE channel (the mask head's out_lay behind gn5) of a channels_last
    x [N, C, H, W], C in {16, 32}: the normalised activation is never written.  -> [N, 1, H, W]; None when not applicable."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and x.shape[1] in (16, 32) and gn.num_channels == x.shape[1] and gn.weight is not None and gn.bias is not None
            and conv.out_channels == 1 and conv.in_channels == x.shape[1] and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and x.shape[0] <= 65535
            and gn.weight.device == x.device and conv.weight.device == x.device and not (x.data_ptr() & 15)):
        return None
    n, c, H, W = x.shape
    hit = getattr(conv, "_tf_c1_taps", None)   # [9, C] tap-major + the bias as a Python float (one synchronising read per weight version)
    key = (conv.weight._version, None if conv.bias is None else conv.bias._version, conv.weight.device)
    if hit is None or hit[0] != key:
        if torch.cuda.is_current_stream_capturing():
            return None
        hit = conv._tf_c1_taps = (key, conv.weight.detach()[0].permute(1, 2, 0).reshape(9, c).contiguous(),
                                  0.0 if conv.bias is None else float(conv.bias.detach()))
    with torch.cuda.device(x.device):
        out = torch.empty((n, 1, H, W), dtype=torch.float32, device=x.device)
        ws = torch.empty(2 * n * gn.num_groups, dtype=torch.float64, device=x.device)
        rc = _cabi.lib().tf_groupnorm_relu_conv3x3_c1_nhwc_f32(x.data_ptr(), gn.weight.data_ptr(), gn.bias.data_ptr(), hit[1].data_ptr(), hit[2],
                                                               out.data_ptr(), ws.data_ptr(), n, H, W, c, gn.num_groups, float(gn.eps),
                                                               _stream(x.device))
    _cabi.check(rc, "tf_groupnorm_relu_conv3x3_c1_nhwc_f32")
    return out


def input_proj_1x1(x, conv, gn):
    """GroupNorm(conv(x)) for the input projections -- 1 x 1 convolutions, and the extra level's 3 x 3 / stride 2 one -- of a
    channels_last fp32 GPU activation x [N, Cin, H, W]; returns [N, Cout, H', W']
    (channels_last) or None when the fused route does not apply (the caller keeps the nn.Sequential)."""
    if not (_input_proj_fused and _split_linear and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and conv.groups == 1):
        return None
    if conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1) and conv.dilation == (1, 1):
        # the extra pyramid level (deformable_detr.py:55-79: Conv2d(2048, hidden, 3, stride 2, padding 1) + GroupNorm)
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        cout, cin = conv.out_channels, conv.in_channels
        hit = getattr(conv, "_tf_wtaps", None)   # persistent [Cout, 9 * Cin] tap-major image: the split pieces are cached on it
        if hit is None or hit[0] != conv.weight._version or hit[1].device != conv.weight.device:
            prev_img = None if hit is None else hit[1]
            hit = (conv.weight._version, conv.weight.detach().permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous())
            inherit_route(hit[1], conv.weight, prev_img)
            conv._tf_wtaps = hit
        y = conv3x3(x, hit[1], conv.bias, False, 2)
        if y is None:
            return None
        n, _, ho, wo = y.shape
        z2 = groupnorm_nhwc(y.permute(0, 2, 3, 1).reshape(n * ho * wo, cout), n, gn)
        return None if z2 is None else z2.view(n, ho, wo, cout).permute(0, 3, 1, 2)
    if not (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)):
        return None
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    n, cin, h, w = x.shape
    cout = conv.out_channels
    hit = getattr(conv, "_tf_w2d", None)   # persistent [Cout, Cin] view: the split pieces are cached on it
    if hit is None or hit[0] != conv.weight._version or hit[1].device != conv.weight.device:
        prev_img = None if hit is None else hit[1]
        hit = (conv.weight._version, conv.weight.detach().reshape(cout, cin).contiguous())
        inherit_route(hit[1], conv.weight, prev_img)
        if hit[1].is_cuda and hit[1].data_ptr() != conv.weight.data_ptr():   # a copy made on this stream: publish for the others
            _publish_barrier(hit[1].device)
        conv._tf_w2d = hit
    y2 = None
    if conv1x1_wants_split_k(n * h * w, cin, cout):   # the coarse levels: 1050 / 4200 pixels under K = 2048 / 1024
        y = conv3x3(x, hit[1], conv.bias, False, 1)
        y2 = None if y is None else y.permute(0, 2, 3, 1)
    if y2 is None:
        y2 = linear(x.permute(0, 2, 3, 1).reshape(n * h * w, cin), hit[1], conv.bias)
    if y2 is None:
        return None
    z2 = groupnorm_nhwc(y2.reshape(n * h * w, cout), n, gn)
    if z2 is None:
        return None
    return z2.view(n, h, w, cout).permute(0, 3, 1, 2)


# DEFAULT since round 3 (TF_BOX_REFINE_FUSED=0 / set_box_refine_fused(False) switches it off; +5.1 % frames/s on its own):
# the decoder's iterative box refinement in one launch
# (tf_box_refine_f32) instead of ~12 element-wise ATen launches per layer on [queries, 4] tensors.
_box_refine_fused = os.environ.get("TF_BOX_REFINE_FUSED", "1") != "0"


def set_box_refine_fused(on):
    global _box_refine_fused
    prev, _box_refine_fused = _box_refine_fused, bool(on)
    return prev


def box_refine(delta, reference_points):
    """sigmoid(delta + inverse_sigmoid(reference_points)) (4-d references) / the 2-d variant of
    deformable_transformer.py:331-343; delta [..., 4], reference_points [..., 2 or 4].  None when not applicable."""
    if not (_box_refine_fused and delta.is_cuda and delta.dtype == torch.float32 and reference_points.dtype == torch.float32
            and delta.shape[-1] == 4 and reference_points.shape[-1] in (2, 4)
            and delta.shape[:-1] == reference_points.shape[:-1] and delta.is_contiguous()
            and reference_points.is_contiguous() and reference_points.device == delta.device and delta.numel() > 0):
        return None
    with torch.cuda.device(delta.device):
        out = torch.empty_like(delta)
        rc = _cabi.lib().tf_box_refine_f32(delta.data_ptr(), reference_points.data_ptr(), out.data_ptr(), delta.numel() // 4,
                                           reference_points.shape[-1], 1e-5, _stream(delta.device))
    _cabi.check(rc, "tf_box_refine_f32")
    return out


# DEFAULT since round 6 (TF_POSTPROCESS_FUSED=0 / set_postprocess_fused(False): the post-processor module + clip + cat, ~17
# element-wise ATen launches on [queries, 4] tensors outside the HIP graph, every frame): what the tracker's association reads
# -- scaled, clipped xyxy boxes, best score, label -- in ONE launch (tf_postprocess_pack_f32).
_postprocess_fused = os.environ.get("TF_POSTPROCESS_FUSED", "1") != "0"


def set_postprocess_fused(on):
    global _postprocess_fused
    prev, _postprocess_fused = _postprocess_fused, bool(on)
    return prev


def postprocess_pack(logits, boxes, img_h, img_w, clip):
    """[Q, 6] = (x0, y0, x1, y1, score, label) of DeformablePostProcess + clip_boxes_to_image for ONE image: logits [Q, C],
    boxes [Q, 4] cxcywh in [0, 1]; the arithmetic of the separate ATen kernels, operation by operation.  None when not applicable."""
    if not (_postprocess_fused and logits.is_cuda and logits.dtype == torch.float32 and boxes.dtype == torch.float32
            and logits.dim() == 2 and boxes.dim() == 2 and boxes.shape == (logits.shape[0], 4) and logits.shape[0] > 0
            and logits.shape[1] > 0 and logits.is_contiguous() and boxes.is_contiguous() and boxes.device == logits.device
            and boxes.data_ptr() % 16 == 0):
        return None
    with torch.cuda.device(logits.device):
        out = torch.empty((logits.shape[0], 6), dtype=torch.float32, device=logits.device)
        rc = _cabi.lib().tf_postprocess_pack_f32(logits.data_ptr(), boxes.data_ptr(), out.data_ptr(), logits.shape[0], logits.shape[1],
                                                 float(img_h), float(img_w), 1 if clip else 0, _stream(logits.device))
    _cabi.check(rc, "tf_postprocess_pack_f32")
    return out


def module_linear(module, x, inference):
    """module(x) for an nn.Linear: the split product on the GPU inference path when enabled, else the module."""
    if inference and _split_linear:
        y = linear(x, module.weight, module.bias)
        if y is not None:
            return y
    return module(x)


# ---------------------------------------------------------------------------------------------------------
def mha_core(qk, v, num_heads, key_padding_mask=None):
    """softmax(q k^T / sqrt(d)) v for the decoder's query self-attention (include/tf_fused.h: tf_mha_core_f32).
    qk [N, L, 2E]: the shared q | k projection of one GEMM; v [N, L, E]; key_padding_mask [N, L] bool (True = ignore)
    or None.  Returns [N, L, E], or None when the kernel does not apply (the caller keeps torch's SDPA)."""
    if os.environ.get("TF_NO_MHA") == "1":   # debugging aid: keep torch's SDPA
        return None
    if not (qk.is_cuda and qk.dtype == torch.float32 and v.dtype == torch.float32 and v.device == qk.device
            and qk.dim() == 3 and v.dim() == 3 and qk.is_contiguous() and v.is_contiguous()):
        return None
    n, length, e2 = qk.shape
    e = e2 // 2
    if e2 != 2 * e or v.shape != (n, length, e) or e % num_heads:
        return None
    d = e // num_heads
    if d not in (16, 32, 36, 64) or e % 4 or length > 2300 or (qk.data_ptr() | v.data_ptr()) & 15 or (e * 4) % 16:
        return None
    mask_ptr = 0
    if key_padding_mask is not None:
        if key_padding_mask.shape != (n, length) or key_padding_mask.device != qk.device:
            return None
        key_padding_mask = key_padding_mask.to(torch.uint8).contiguous()
        mask_ptr = key_padding_mask.data_ptr()
    with torch.cuda.device(qk.device):
        out = torch.empty((n, length, e), dtype=torch.float32, device=qk.device)
        q_ptr = qk.data_ptr()
        rc = _cabi.lib().tf_mha_core_f32(q_ptr, q_ptr + e * 4, v.data_ptr(), out.data_ptr(), mask_ptr, n, length, length,
                                         num_heads, d, e2, e2, e, e, float(d) ** -0.5, _stream(qk.device))
    if rc == -2:   # TF_MSDA_ERR_BAD_DIMS: more keys than the score tile has LDS for -> the caller keeps torch's SDPA
        return None
    _cabi.check(rc, "tf_mha_core_f32")
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# The fp16 split product's activation range (csrc/split_product.h): activations are scaled by 2^-4 into fp16, so |x| up to
# 65504 * 16 = 1.05e6 is representable; beyond it the hi piece is inf and the output row is NaN -- loudly: every ReLU epilogue
# propagates NaN (v < 0 ? 0 : v, as torch.relu does), none maps it to zero.  The reference's fp32 arithmetic has no such limit,
# so three tools make the contract explicit instead of silent:
#   * set_check_finite(True) / TF_SPLIT_CHECK_FINITE=1 -- debug mode: every split-product call checks its result and raises a
#     FloatingPointError naming the operation (one host synchronisation per call: not for timed runs);
#   * audit_activation_range() -- a context manager: run the model once under it on representative input; it records the
#     largest |activation| every split-product layer was given (and, for the fused feed-forward block, its hidden
#     activation) and, on exit, routes each layer that came within `safety` of the limit through the SIX-TERM bf16 product
#     (all 24 significand bits, fp32's exponent range) for the rest of the process; the report lists them;
#   * route_six_terms(weight) -- the same routing by hand.
# With no layer routed, no audit running and the check off the wrappers cost one comparison per call.
import contextlib
import functools

F16_ACTIVATION_LIMIT = 65504.0 * 16.0
_check_finite = os.environ.get("TF_SPLIT_CHECK_FINITE", "0") not in ("", "0")
_n_six_term_routes = 0        # weights marked `_tf_six_terms` (their layers run the six-term product while the default is the fp16 one)
_range_audit = None           # {(operation, ids of the weights): [weights, shapes, largest |activation|]} while audit_activation_range() runs
_route_epoch = 0              # bumped whenever a route / the check / an audit changes what a forward enqueues: GraphedDetector
                              # drops the HIP graphs it captured under another epoch (kernels are baked into a graph)

# HIP graphs (graphed.GraphedDetector, the default path of bench.py and dist_utils.track_sequences): a replay calls none of the
# wrappers below, so (1) an audit or the finite check sees nothing inside a graph -- GraphedDetector therefore runs EAGERLY while
# either is active (debug_checks_active()), and a capture attempted with one of them on raises instead of synchronising inside
# the capture; (2) a route added after a graph was captured would not reach it -- route_epoch() is part of what a graph was
# captured under, a change drops the graphs and they are captured again with the routed kernels.  Run the audit BEFORE timing.


def route_epoch():
    return _route_epoch


def debug_checks_active():
    """An activation-range audit is running or the finite check is on: forwards must run eagerly (not from a captured graph)."""
    return _check_finite or _range_audit is not None


def set_check_finite(on):
    global _check_finite, _route_epoch
    prev, _check_finite = _check_finite, bool(on)
    if prev != _check_finite:
        _route_epoch += 1
    return prev


def _routed(weight):
    return getattr(weight, "_tf_six_terms", False)


def route_six_terms(weight, on=True):
    """Route every split product of this weight tensor through the six-term bf16 product (while the default is the fp16 one).
    The mark lives ON THE TENSOR OBJECT (like the cached pieces: an address can be freed and handed to another weight) and is
    inherited by the images derived from it (inherit_route: a convolution's tap-major / 2-d image, re-created when the weight
    changes)."""
    global _n_six_term_routes, _route_epoch
    if bool(_routed(weight)) != bool(on):
        weight._tf_six_terms = bool(on)
        _n_six_term_routes += 1 if on else -1
        _route_epoch += 1


def inherit_route(new, *sources):
    """`new` is an image derived from `sources` (the parameter, the image it replaces): it runs six terms if one of them does."""
    if any(src is not None and _routed(src) for src in sources) and not _routed(new):
        route_six_terms(new)
    return new


def six_term_routes():
    return _n_six_term_routes


@contextlib.contextmanager
def audit_activation_range(route=True, safety=4.0):
    """-> report dict, filled on exit: {"limit", "largest", "layers": [(operation, weight shape, largest |activation|)...],
    "routed": n}.  Layers whose activations came within `safety` of the fp16 product's limit are routed to six terms.
    Forwards under the audit run eagerly (GraphedDetector steps aside); graphs captured before it are dropped if it routes."""
    global _range_audit, _route_epoch
    report = {"limit": F16_ACTIVATION_LIMIT, "safety": safety, "largest": 0.0, "layers": [], "routed": 0}
    prev, _range_audit = _range_audit, {}
    _route_epoch += 1
    try:
        yield report
    finally:
        seen, _range_audit = _range_audit, prev
        _route_epoch += 1
        for (name, _ids), (ws, shapes, amax) in sorted(seen.items(), key=lambda kv: -kv[1][2]):
            report["layers"].append((name, shapes, amax))
            report["largest"] = max(report["largest"], amax)
            if route and not (amax * safety < F16_ACTIVATION_LIMIT):   # also catches NaN / inf
                for w in ws:
                    route_six_terms(w)
                report["routed"] += 1


def _amax(t):
    return float(t.detach().abs().max()) if t is not None and t.numel() else 0.0


def _ranged(name, weights_of, acts_of):
    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            if _split_terms != 16 or not (_n_six_term_routes or _check_finite or _range_audit is not None):
                return fn(*a, **k)
            ws = [w for w in weights_of(*a, **k) if w is not None]
            checks = _check_finite or _range_audit is not None
            if checks and ws and ws[0].is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("fused.%s: an activation-range audit / the finite check (set_check_finite, TF_SPLIT_CHECK_FINITE) "
                                   "reads results on the host and cannot run while a HIP graph is being captured -- run the model "
                                   "eagerly under them (GraphedDetector does so by itself), then capture" % name)
            route6 = any(_routed(w) for w in ws)
            if route6:
                _tls.terms = 6     # this thread's calls only: another tracker thread of the same model keeps its own arithmetic
            try:
                y = fn(*a, **k)
            finally:
                if route6:
                    _tls.terms = None
            if y is None:
                return y
            if _range_audit is not None:
                key = (name, tuple(id(w) for w in ws))
                rec = _range_audit.setdefault(key, [ws, tuple(tuple(w.shape) for w in ws), 0.0])   # (holds the weights: ids stay unique)
                rec[2] = max(rec[2], max(_amax(t) for t in acts_of(*a, **k)))
            if _check_finite and not bool(torch.isfinite(y).all()):
                raise FloatingPointError(
                    "%s produced a non-finite result (largest |input| %.3g; the fp16 split product represents activations up to "
                    "%.3g -- fused.audit_activation_range() routes such layers through the six-term product, "
                    "fused.set_split_terms(6) all of them)" % (name, max(_amax(t) for t in acts_of(*a, **k)), F16_ACTIVATION_LIMIT))
            return y
        return wrapped
    return deco


def _ffn_acts(x, linear1, linear2, norm=None, residual=None):
    hidden = F.relu(F.linear(x, linear1.weight, linear1.bias)) if _range_audit is not None else None   # audit only: the hidden activation
    return (x, hidden)


linear = _ranged("linear", lambda x, weight, *a, **k: (weight,), lambda x, *a, **k: (x,))(linear)
linear_add = _ranged("linear_add", lambda x, x2, weight, *a, **k: (weight,), lambda x, x2, *a, **k: (x + x2,))(linear_add)
ffn = _ranged("ffn", lambda x, linear1, linear2, *a, **k: (linear1.weight, linear2.weight), _ffn_acts)(ffn)
linear_residual_norm = _ranged("linear_residual_norm", lambda x, linear, *a, **k: (linear.weight,), lambda x, *a, **k: (x,))(linear_residual_norm)
stem_conv = _ranged("stem_conv", lambda x, weight, *a, **k: (weight,), lambda x, *a, **k: (x,))(stem_conv)
conv3x3 = _ranged("conv3x3", lambda x, w_taps, *a, **k: (w_taps,), lambda x, *a, **k: (x,))(conv3x3)
