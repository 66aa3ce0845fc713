// trackformer_amd/csrc/msda_host.cpp -- the operator for HOST tensors (include/tf_msda.h: tf_msda_*_host_*).
//
// The reference dispatches on the device of `value` and has nothing behind the CPU branch:
// ops/src/ms_deform_attn.h:27,48 -> AT_ERROR("Not implemented on the CPU"), ops/src/cpu/ms_deform_attn_cpu.cpp is a stub.
// SURVEY.md section 8(b) asks for a real CPU path in its place.  This is that path and nothing else: it is reached only
// for tensors that already live in host memory (CPU-side model tests, small debugging runs, the reference's own
// MSDeformAttnFunction called on CPU tensors through the drop-in module); device tensors never come here -- the HIP
// entry points have no host fallback -- and a missing libtf_msda.so fails for both.
//
// Arithmetic: SURVEY.md Appendix A, i.e. ms_deform_im2col_cuda.cuh:227-237 (pixel = loc * size - 0.5, in range iff
// -1 < pixel < size), :24-67 (bilinear taps, zero padding), :69-163 (gradients).  Work is split over host threads by
// (batch, head): a head's slice of grad_value belongs to exactly one thread, so the backward needs no atomics and its
// result is deterministic.
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

#include "tf_msda.h"

namespace {

struct Dims {
    int N, S, M, D, L, Lq, P;
};

bool dims_ok(const Dims &d, const int64_t *shapes, int64_t *starts)
{
    if (d.N <= 0 || d.S <= 0 || d.M <= 0 || d.D <= 0 || d.L <= 0 || d.Lq <= 0 || d.P <= 0 || d.L > TF_MSDA_MAX_LEVELS) return false;
    int64_t sum = 0;
    for (int l = 0; l < d.L; ++l) {
        if (shapes[2 * l] <= 0 || shapes[2 * l + 1] <= 0) return false;
        starts[l] = sum;
        sum += shapes[2 * l] * shapes[2 * l + 1];
    }
    return sum == d.S;
}

// fn(item) for item in [0, n): contiguous blocks over at most hardware_concurrency() threads
template <class F>
void parallel_items(int64_t n, F fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    int64_t nt = std::max<int64_t>(1, std::min<int64_t>(n, hw ? hw : 1));
    if (nt == 1) {
        for (int64_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)nt);
    for (int64_t t = 0; t < nt; ++t)
        pool.emplace_back([=] {
            const int64_t a = n * t / nt, b = n * (t + 1) / nt;
            for (int64_t i = a; i < b; ++i) fn(i);
        });
    for (auto &th : pool) th.join();
}

// One sampling point: the (up to) four rows it touches and their bilinear weights.
template <class T>
struct Taps {
    bool in;
    int64_t row[4];   // row index inside the level, -1 = outside (zero padding)
    T w[4];           // (1-ly)(1-lx), (1-ly) lx, ly (1-lx), ly lx
    T ly, lx;
};

template <class T>
Taps<T> make_taps(T loc_x, T loc_y, int64_t H, int64_t W)
{
    Taps<T> t;
    const T y = loc_y * (T)H - (T)0.5, x = loc_x * (T)W - (T)0.5;   // cuh:227-228
    t.in = y > (T)-1 && x > (T)-1 && y < (T)H && x < (T)W;          // cuh:229
    if (!t.in) return t;
    const T yf = std::floor(y), xf = std::floor(x);
    const int64_t y0 = (int64_t)yf, x0 = (int64_t)xf, y1 = y0 + 1, x1 = x0 + 1;
    t.ly = y - yf;
    t.lx = x - xf;
    const T hy = (T)1 - t.ly, hx = (T)1 - t.lx;
    t.row[0] = (y0 >= 0 && x0 >= 0) ? y0 * W + x0 : -1;             // cuh:44-62
    t.row[1] = (y0 >= 0 && x1 <= W - 1) ? y0 * W + x1 : -1;
    t.row[2] = (y1 <= H - 1 && x0 >= 0) ? y1 * W + x0 : -1;
    t.row[3] = (y1 <= H - 1 && x1 <= W - 1) ? y1 * W + x1 : -1;
    t.w[0] = hy * hx;
    t.w[1] = hy * t.lx;
    t.w[2] = t.ly * hx;
    t.w[3] = t.ly * t.lx;
    return t;
}

template <class T>
int forward_host(const T *value, const int64_t *shapes, const T *loc, const T *attn, T *out, const Dims d)
{
    if (!value || !shapes || !loc || !attn || !out) return TF_MSDA_ERR_NULL_POINTER;
    int64_t starts[TF_MSDA_MAX_LEVELS];
    if (d.L > TF_MSDA_MAX_LEVELS || d.L <= 0) return TF_MSDA_ERR_BAD_DIMS;
    if (!dims_ok(d, shapes, starts)) {
        int64_t sum = 0;
        for (int l = 0; l < d.L; ++l) sum += shapes[2 * l] * shapes[2 * l + 1];
        return (d.N > 0 && d.S > 0 && d.M > 0 && d.D > 0 && d.Lq > 0 && d.P > 0 && sum != d.S) ? TF_MSDA_ERR_SHAPE_SUM : TF_MSDA_ERR_BAD_DIMS;
    }
    const int64_t LP = (int64_t)d.L * d.P;
    parallel_items((int64_t)d.N * d.Lq, [&](int64_t nq) {
        const int64_t n = nq / d.Lq;
        for (int m = 0; m < d.M; ++m) {
            T *o = out + (nq * d.M + m) * d.D;
            for (int c = 0; c < d.D; ++c) o[c] = (T)0;
            const T *pl = loc + (nq * d.M + m) * LP * 2, *pa = attn + (nq * d.M + m) * LP;
            for (int l = 0; l < d.L; ++l) {
                const int64_t H = shapes[2 * l], W = shapes[2 * l + 1];
                const T *lvl = value + ((n * d.S + starts[l]) * d.M + m) * d.D;
                for (int p = 0; p < d.P; ++p) {
                    const Taps<T> t = make_taps<T>(pl[(l * d.P + p) * 2], pl[(l * d.P + p) * 2 + 1], H, W);
                    if (!t.in) continue;
                    const T a = pa[l * d.P + p];
                    for (int k = 0; k < 4; ++k) {
                        if (t.row[k] < 0) continue;
                        const T *v = lvl + t.row[k] * d.M * d.D;
                        const T wk = t.w[k] * a;
                        for (int c = 0; c < d.D; ++c) o[c] += wk * v[c];
                    }
                }
            }
        }
    });
    return TF_MSDA_OK;
}

template <class T>
int backward_host(const T *value, const int64_t *shapes, const T *loc, const T *attn, const T *grad_out, T *grad_value,
                  T *grad_loc, T *grad_attn, const Dims d)
{
    if (!value || !shapes || !loc || !attn || !grad_out || !grad_value || !grad_loc || !grad_attn) return TF_MSDA_ERR_NULL_POINTER;
    int64_t starts[TF_MSDA_MAX_LEVELS];
    if (d.L > TF_MSDA_MAX_LEVELS || d.L <= 0) return TF_MSDA_ERR_BAD_DIMS;
    if (!dims_ok(d, shapes, starts)) {
        int64_t sum = 0;
        for (int l = 0; l < d.L; ++l) sum += shapes[2 * l] * shapes[2 * l + 1];
        return (d.N > 0 && d.S > 0 && d.M > 0 && d.D > 0 && d.Lq > 0 && d.P > 0 && sum != d.S) ? TF_MSDA_ERR_SHAPE_SUM : TF_MSDA_ERR_BAD_DIMS;
    }
    const int64_t LP = (int64_t)d.L * d.P;
    parallel_items((int64_t)d.N * d.M, [&](int64_t nm) {
        const int64_t n = nm / d.M;
        const int m = (int)(nm % d.M);
        for (int64_t s = 0; s < d.S; ++s) {   // this thread's slice of grad_value (cu:119-121: zeros)
            T *g = grad_value + ((n * d.S + s) * d.M + m) * d.D;
            for (int c = 0; c < d.D; ++c) g[c] = (T)0;
        }
        for (int64_t q = 0; q < d.Lq; ++q) {
            const int64_t pair = (n * d.Lq + q) * d.M + m;
            const T *go = grad_out + pair * d.D;
            const T *pl = loc + pair * LP * 2, *pa = attn + pair * LP;
            T *gl = grad_loc + pair * LP * 2, *ga = grad_attn + pair * LP;
            for (int l = 0; l < d.L; ++l) {
                const int64_t H = shapes[2 * l], W = shapes[2 * l + 1];
                const int64_t base = ((n * d.S + starts[l]) * d.M + m) * d.D;
                for (int p = 0; p < d.P; ++p) {
                    const int64_t i = (int64_t)l * d.P + p;
                    gl[2 * i] = gl[2 * i + 1] = ga[i] = (T)0;
                    const Taps<T> t = make_taps<T>(pl[2 * i], pl[2 * i + 1], H, W);
                    if (!t.in) continue;
                    const T a = pa[i], hy = (T)1 - t.ly, hx = (T)1 - t.lx;
                    // d(sample)/dy and /dx per channel are linear in the four rows (cuh:104-147): with s_k = <grad_out, row k>
                    T sk[4] = {(T)0, (T)0, (T)0, (T)0};
                    for (int k = 0; k < 4; ++k) {
                        if (t.row[k] < 0) continue;
                        const T *v = value + base + t.row[k] * d.M * d.D;
                        T *g = grad_value + base + t.row[k] * d.M * d.D;
                        const T wk = t.w[k] * a;
                        T acc = (T)0;
                        for (int c = 0; c < d.D; ++c) {
                            acc += go[c] * v[c];
                            g[c] += wk * go[c];   // cuh:110,121,132,143: w_k * (grad_out * attn)
                        }
                        sk[k] = acc;
                    }
                    ga[i] = t.w[0] * sk[0] + t.w[1] * sk[1] + t.w[2] * sk[2] + t.w[3] * sk[3];                       // cuh:150-151
                    gl[2 * i] = (T)W * a * (hy * (sk[1] - sk[0]) + t.ly * (sk[3] - sk[2]));                         // d/dx, cuh:152
                    gl[2 * i + 1] = (T)H * a * (hx * (sk[2] - sk[0]) + t.lx * (sk[3] - sk[1]));                     // d/dy, cuh:153
                }
            }
        }
    });
    return TF_MSDA_OK;
}

}  // namespace

extern "C" {

int tf_msda_forward_host_f32(const float *value, const int64_t *shapes_hw, const float *loc, const float *attn, float *out, int N,
                             int S, int M, int D, int L, int Lq, int P)
{
    return forward_host<float>(value, shapes_hw, loc, attn, out, Dims{N, S, M, D, L, Lq, P});
}
int tf_msda_forward_host_f64(const double *value, const int64_t *shapes_hw, const double *loc, const double *attn, double *out,
                             int N, int S, int M, int D, int L, int Lq, int P)
{
    return forward_host<double>(value, shapes_hw, loc, attn, out, Dims{N, S, M, D, L, Lq, P});
}
int tf_msda_backward_host_f32(const float *value, const int64_t *shapes_hw, const float *loc, const float *attn,
                              const float *grad_out, float *grad_value, float *grad_loc, float *grad_attn, int N, int S, int M,
                              int D, int L, int Lq, int P)
{
    return backward_host<float>(value, shapes_hw, loc, attn, grad_out, grad_value, grad_loc, grad_attn, Dims{N, S, M, D, L, Lq, P});
}
int tf_msda_backward_host_f64(const double *value, const int64_t *shapes_hw, const double *loc, const double *attn,
                              const double *grad_out, double *grad_value, double *grad_loc, double *grad_attn, int N, int S,
                              int M, int D, int L, int Lq, int P)
{
    return backward_host<double>(value, shapes_hw, loc, attn, grad_out, grad_value, grad_loc, grad_attn, Dims{N, S, M, D, L, Lq, P});
}

}  // extern "C"

// ---- greedy non-maximum suppression on host boxes (torchvision.ops.nms semantics; reference call sites tracker.py:399,495).
// The association leg of Tracker.step runs it twice per frame on a few hundred host boxes; as a K x K IoU matrix plus a
// Python sweep it cost ~0.35 ms per call.  Same arithmetic as box_ops.box_iou in the same order (fp32, no contraction),
// same visiting order as torch.sort(descending=True, stable=True): the kept set is identical.
#pragma STDC FP_CONTRACT OFF
extern "C" int tf_nms_host_f32(const float *boxes, const float *scores, int n, float iou_threshold, int64_t *keep, int *n_keep)
{
    if (!boxes || !scores || !keep || !n_keep) return TF_MSDA_ERR_NULL_POINTER;
    if (n < 0) return TF_MSDA_ERR_BAD_DIMS;
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; ++i) order[(size_t)i] = i;
    // descending by score, ties in input order (tracker.py:493-495 relies on it for its +inf scores); a NaN score sorts in front
    // of everything, as torch.sort(descending=True) places it -- `a > b` alone is not a strict weak ordering with NaNs (undefined
    // behaviour in std::stable_sort)
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        const float sa = scores[a], sb = scores[b];
        const bool na = sa != sa, nb = sb != sb;
        return na ? !nb : (!nb && sa > sb);
    });
    std::vector<float> area((size_t)n);
    for (int i = 0; i < n; ++i) area[(size_t)i] = (boxes[4 * i + 2] - boxes[4 * i]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    std::vector<char> dead((size_t)n, 0);
    int kept = 0;
    for (int a = 0; a < n; ++a) {
        const int i = order[(size_t)a];
        if (dead[(size_t)a]) continue;
        keep[kept++] = i;
        const float *bi = boxes + 4 * i;
        for (int b = a + 1; b < n; ++b) {
            if (dead[(size_t)b]) continue;
            const int j = order[(size_t)b];
            const float *bj = boxes + 4 * j;
            const float ltx = std::max(bi[0], bj[0]), lty = std::max(bi[1], bj[1]);
            const float rbx = std::min(bi[2], bj[2]), rby = std::min(bi[3], bj[3]);
            const float w = std::max(rbx - ltx, 0.f), h = std::max(rby - lty, 0.f);
            const float inter = w * h;
            const float uni = area[(size_t)i] + area[(size_t)j] - inter;
            const float iou = inter / uni;
            if (iou > iou_threshold) dead[(size_t)b] = 1;
        }
    }
    *n_keep = kept;
    return TF_MSDA_OK;
}
