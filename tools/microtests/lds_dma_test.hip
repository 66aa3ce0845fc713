// Micro-test: semantics of buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950.
// Each lane supplies its own byte offset; where does its 16-byte piece land in LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__global__ void k(const float *src, unsigned nbytes, const unsigned *offs, float *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nbytes, 0x00020000);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) ((float *)smem)[i] = -1.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    // wave w writes its 1 KiB at LDS byte offset 2048*w (+ instruction offset 0)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(smem + 2048 * wave),
                                         16, offs[threadIdx.x], 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = ((float *)smem)[i];
}

int main()
{
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    std::vector<unsigned> ho(128);
    for (int i = 0; i < 128; ++i) ho[i] = (unsigned)(((i * 37) % 200) * 16);   // scattered 16-B pieces
    ho[5] = 0xFFFFFFF0u;                                                       // out of range -> zeros?
    float *d, *o; unsigned *dofs;
    hipMalloc(&d, n * 4); hipMalloc(&o, 1024 * 4); hipMalloc(&dofs, 128 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dofs, ho.data(), 128 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 4096 * 2, 0, d, (unsigned)(n * 4), dofs, o);
    std::vector<float> r(1024);
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 2; ++w)
        for (int lane = 0; lane < 64; ++lane) {
            const int t = w * 64 + lane;
            for (int j = 0; j < 4; ++j) {
                const float got = r[(2048 * w) / 4 + lane * 4 + j];
                const float exp = (t == 5) ? 0.f : (float)(ho[t] / 4 + j);
                if (got != exp) { if (bad < 8) printf("wave %d lane %d j %d got %g exp %g\n", w, lane, j, got, exp); ++bad; }
            }
        }
    printf("untouched gap value (expect -1): %g\n", r[1024 / 4 + 3]);
    printf(bad ? "MISMATCH %d\n" : "LDS-DMA layout OK: lane i -> lds_base + 16*i, OOB -> 0\n", bad);
    return bad != 0;
}
