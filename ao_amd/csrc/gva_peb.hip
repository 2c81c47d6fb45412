// ao_amd/csrc/gva_peb.hip -- the grouped positional-bias projection of the fused GVA (gfx950).
//
//   out[n, g*I+i] = out_v[n, g*I+i] + sum_c' A[n,g,c'] * Wp2[g*I+i, c'] + bp2[g*I+i] * sw[n,g]
//
// i.e. linear_p_bias[3] applied AFTER the softmax-weighted sum over neighbours (see ao_amd/ptv2/gva.py).
// It is a batch of G thin GEMMs (N x C') x (C' x I) with I = C/G = 8 output columns each -- a shape
// rocBLAS serves poorly (measured 353 us per call at N = 120k, profiles/r01_fused_v1_*).  Here a
// workgroup streams a tile of T = 256/I points of A[g] through LDS once per group (coalesced,
// read-once: the kernel is HBM-bound on A) and each thread owns one output element.
#include <algorithm>

#include "gva_common.h"

namespace gva {

template <int I>
__global__ __launch_bounds__(TPB) void peb_fwd_kernel(int n, int c, int g, const float *__restrict__ A,
                                                      const float *__restrict__ Wp2, const float *__restrict__ bp2,
                                                      const float *__restrict__ sw, const float *__restrict__ out_v,
                                                      float *__restrict__ out) {
    constexpr int T = TPB / I;  // points per tile
    extern __shared__ float4 lds4[];
    float *sA = (float *)lds4;            // [T][c + 4]  (padded rows: threads of different points hit different banks)
    const int ldA = c + 4;
    float *sW = sA + (size_t)T * ldA;      // [I][c]
    const int t = threadIdx.x / I, i = threadIdx.x - t * I;
    const int ntiles = (n + T - 1) / T;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n0 = tile * T;
        const int cnt = (n - n0) < T ? (n - n0) : T;
        for (int gi = 0; gi < g; ++gi) {
            __syncthreads();
            for (int e = threadIdx.x; e < cnt * (c / 4); e += TPB) {
                const int r = e / (c / 4), q = e - r * (c / 4);
                *(float4 *)(sA + (size_t)r * ldA + 4 * q) = *(const float4 *)(A + ((size_t)(n0 + r) * g + gi) * c + 4 * q);
            }
            const float4 *wsrc = (const float4 *)(Wp2 + (size_t)gi * I * c);
            for (int e = threadIdx.x; e < I * (c / 4); e += TPB) ((float4 *)sW)[e] = wsrc[e];
            __syncthreads();
            if (t < cnt) {
                const float4 *ar = (const float4 *)(sA + (size_t)t * ldA), *wr = (const float4 *)(sW + (size_t)i * c);
                float acc = 0.f;
                for (int q = 0; q < c / 4; ++q) {
                    const float4 x = ar[q], w = wr[q];
                    acc = __builtin_fmaf(x.x, w.x, acc);
                    acc = __builtin_fmaf(x.y, w.y, acc);
                    acc = __builtin_fmaf(x.z, w.z, acc);
                    acc = __builtin_fmaf(x.w, w.w, acc);
                }
                const size_t o = (size_t)(n0 + t) * c + gi * I + i;
                out[o] = out_v[o] + acc + bp2[gi * I + i] * sw[(size_t)(n0 + t) * g + gi];
            }
        }
    }
}

// gA[n,g,c'] = sum_i gO[n, g*I+i] * Wp2[g*I+i, c'] ;  g_sw[n,g] = sum_i gO[n, g*I+i] * bp2[g*I+i]
template <int I>
__global__ __launch_bounds__(TPB) void peb_bwd_kernel(int n, int c, int g, const float *__restrict__ gO,
                                                      const float *__restrict__ Wp2, const float *__restrict__ bp2,
                                                      float *__restrict__ gA, float *__restrict__ g_sw) {
    const long long total = (long long)g * n * (c / 4);
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const int q = (int)(e % (c / 4));
        const long long ng = e / (c / 4);
        const int gi = (int)(ng % g), nn = (int)(ng / g);
        const float *go = gO + (size_t)nn * c + gi * I;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < I; ++i) {
            const float s = go[i];
            const float4 w = *(const float4 *)(Wp2 + (size_t)(gi * I + i) * c + 4 * q);
            acc.x = __builtin_fmaf(s, w.x, acc.x);
            acc.y = __builtin_fmaf(s, w.y, acc.y);
            acc.z = __builtin_fmaf(s, w.z, acc.z);
            acc.w = __builtin_fmaf(s, w.w, acc.w);
        }
        ((float4 *)gA)[e] = acc;
        if (q == 0) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < I; ++i) t = __builtin_fmaf(go[i], bp2[gi * I + i], t);
            g_sw[(size_t)nn * g + gi] = t;
        }
    }
}

}  // namespace gva

using namespace gva;

#define PEB_DISPATCH_I(i, CALL)        \
    switch (i) {                       \
        case 2: { CALL(2); break; }    \
        case 4: { CALL(4); break; }    \
        case 8: { CALL(8); break; }    \
        case 16: { CALL(16); break; }  \
        default: return PTV2_ERR_ARG;  \
    }

extern "C" int gva_peb_forward_hip_launcher(int n, int c, int g, const float *A, const float *Wp2, const float *bp2,
                                            const float *sw, const float *out_v, float *out, void *stream) {
    if (n < 0 || c < 4 || g < 1 || c % g != 0 || c % 4 != 0) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const int I = c / g;
    if (TPB % I != 0) return PTV2_ERR_ARG;
    const int T = TPB / I;
    const size_t lds = sizeof(float) * ((size_t)T * (c + 4) + (size_t)I * c);
    if (lds > 160 * 1024) return PTV2_ERR_ARG;
    const int ntiles = (n + T - 1) / T;
    const int nblk = ntiles < 256 * 4 ? ntiles : 256 * 4;
#define CALL(II)                                                                                                      \
    if (lds > 32 * 1024)                                                                                              \
        (void)hipFuncSetAttribute((const void *)peb_fwd_kernel<II>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(peb_fwd_kernel<II>, dim3(nblk), dim3(TPB), lds, (hipStream_t)stream, n, c, g, A, Wp2, bp2, sw, out_v, out)
    PEB_DISPATCH_I(I, CALL)
#undef CALL
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int gva_peb_backward_hip_launcher(int n, int c, int g, const float *g_out, const float *Wp2,
                                             const float *bp2, float *g_A, float *g_sw, void *stream) {
    if (n < 0 || c < 4 || g < 1 || c % g != 0 || c % 4 != 0) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const int I = c / g;
    const long long total = (long long)g * n * (c / 4);
    const int nblk = (int)std::min<long long>((total + TPB - 1) / TPB, 256 * 16);
#define CALL(II) \
    hipLaunchKernelGGL(peb_bwd_kernel<II>, dim3(nblk), dim3(TPB), 0, (hipStream_t)stream, n, c, g, g_out, Wp2, bp2, g_A, g_sw)
    PEB_DISPATCH_I(I, CALL)
#undef CALL
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
