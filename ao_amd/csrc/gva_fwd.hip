// ao_amd/csrc/gva_fwd.hip -- forward stages of the fused grouped vector attention (gfx950).
// Math and notation: ao_amd/ptv2/gva.py (module docstring); reference op sequence:
// pointcept/models/point_transformer_v2/point_transformer_v2m2_base.py:103-129.
//
//   gva_pos_stats       sum pos, sum pos pos^T over all N*K neighbour slots (BN_p closed form)
//   gva_logits_forward  W1[n,s,:] = kW[idx] - qW[n] + ReLU(pos a^T + b) M + cW, plus per-channel
//                       sum / sum-of-squares for BN_w.  One lane per neighbour slot, the G logits in
//                       registers; a, b, M are wave-uniform operands (scalar loads, SGPR FMA sources).
//                       HBM traffic per launch: idx + coord + kW/qW gathers in, W1 out -- the
//                       (N,K,C) intermediate of the reference never exists.
// (softmax / aggregation stages: gva_aggregate.hip)
#include "gva_common.h"

namespace gva {

// ------------------------------------------------------------------ pos stats --
__global__ __launch_bounds__(TPB) void pos_stats_kernel(int n, int k, const float *__restrict__ coord,
                                                        const int *__restrict__ idx, float *__restrict__ part) {
    __shared__ float s_w[WPB][9];
    const long long rows = (long long)n * k;
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (long long row = (long long)blockIdx.x * TPB + threadIdx.x; row < rows; row += (long long)gridDim.x * TPB) {
        Rel r = rel_pos(coord, idx, row, (int)(row / k));
        acc[0] += r.x; acc[1] += r.y; acc[2] += r.z;
        acc[3] += r.x * r.x; acc[4] += r.x * r.y; acc[5] += r.x * r.z;
        acc[6] += r.y * r.y; acc[7] += r.y * r.z; acc[8] += r.z * r.z;
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        float v = wave_sum(acc[j]);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        float v = 0.f;
        for (int w = 0; w < WPB; ++w) v += s_w[w][threadIdx.x];
        part[blockIdx.x * 9 + threadIdx.x] = v;
    }
}

struct MapPosStats {  // columns: x y z xx xy xz yy yz zz -> s1[3], symmetric s2[9]
    double *s1, *s2;
    __device__ void operator()(int j, double v) const {
        if (j < 3) { s1[j] = v; return; }
        const int r[6] = {0, 0, 0, 1, 1, 2}, c[6] = {0, 1, 2, 1, 2, 2};
        const int a = r[j - 3], b = c[j - 3];
        s2[a * 3 + b] = v;
        s2[b * 3 + a] = v;
    }
};

// ------------------------------------------------------------- logits forward --
template <int G>
__global__ __launch_bounds__(TPB) void logits_fwd_kernel(int n, int k, int c, const float *__restrict__ kW,
                                                         const float *__restrict__ qW, const float *__restrict__ a,
                                                         const float *__restrict__ b, const float *__restrict__ M,
                                                         const float *__restrict__ cW, const float *__restrict__ coord,
                                                         const int *__restrict__ idx, float *__restrict__ W1,
                                                         float *__restrict__ part) {
    __shared__ float s_w[WPB][2 * G];
    const long long rows = (long long)n * k;
    float t1[G], t2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) t1[g] = t2[g] = 0.f;
    for (long long row = (long long)blockIdx.x * TPB + threadIdx.x; row < rows; row += (long long)gridDim.x * TPB) {
        const int nn = (int)(row / k);
        const Rel r = rel_pos(coord, idx, row, nn);
        float acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float kv = r.src >= 0 ? kW[(long long)r.src * G + g] : 0.f;
            acc[g] = (kv - qW[(long long)nn * G + g]) + cW[g];
        }
        for (int ci = 0; ci < c; ++ci) {  // wave-uniform operands: a, b, M
            const float p = pe_act(a[3 * ci], a[3 * ci + 1], a[3 * ci + 2], b[ci], r.x, r.y, r.z);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_fmaf(p, M[ci * G + g], acc[g]);
        }
        float *o = W1 + row * G;
        if (G % 4 == 0) {
#pragma unroll
            for (int g = 0; g < G; g += 4) *(float4 *)(o + g) = make_float4(acc[g], acc[g + 1], acc[g + 2], acc[g + 3]);
        } else if (G % 2 == 0) {
#pragma unroll
            for (int g = 0; g < G; g += 2) *(float2 *)(o + g) = make_float2(acc[g], acc[g + 1]);
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) o[g] = acc[g];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            t1[g] += acc[g];
            t2[g] = __builtin_fmaf(acc[g], acc[g], t2[g]);
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float v1 = wave_sum(t1[g]), v2 = wave_sum(t2[g]);
        if ((threadIdx.x & 63) == 0) {
            s_w[threadIdx.x >> 6][g] = v1;
            s_w[threadIdx.x >> 6][G + g] = v2;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * G) {
        float v = 0.f;
        for (int w = 0; w < WPB; ++w) v += s_w[w][threadIdx.x];
        part[(size_t)blockIdx.x * 2 * G + threadIdx.x] = v;
    }
}

inline int stage_grid(long long work_items, int per_block) {
    long long b = (work_items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 256 * 8 ? 256 * 8 : b));
}

inline bool pow2(int k) { return k > 0 && (k & (k - 1)) == 0; }

}  // namespace gva

using namespace gva;

extern "C" size_t gva_workspace_bytes(int n, int k, int c, int g) {
    if (n < 0 || k < 1 || c < 1 || g < 1) return 0;
    // [per-block partial sums of the widest stage][one (n,k,g) fp32 row buffer: gWt / w]
    return rows_offset_bytes(c, g) + align_up(sizeof(float) * (size_t)n * k * g) + 1024;
}

extern "C" int gva_pos_stats_hip_launcher(int n, int k, const float *coord, const int *idx, double *s1, double *s2,
                                          void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 0 || k < 1) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < sizeof(float) * 9 * 2048) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    int nblk = stage_grid((long long)n * k, TPB * 4);
    hipLaunchKernelGGL(pos_stats_kernel, dim3(nblk), dim3(TPB), 0, st, n, k, coord, idx, part);
    launch_finalize(st, (const float *)part, nblk, 9, MapPosStats{s1, s2});
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

#define GVA_DISPATCH_G(g, CALL)            \
    switch (g) {                           \
        case 6: { CALL(6); break; }        \
        case 12: { CALL(12); break; }      \
        case 24: { CALL(24); break; }      \
        case 48: { CALL(48); break; }      \
        case 64: { CALL(64); break; }      \
        default: return PTV2_ERR_ARG;      \
    }

extern "C" int gva_logits_forward_hip_launcher(int n, int k, int c, int g, const float *kW, const float *qW,
                                               const float *a, const float *b, const float *M, const float *cW,
                                               const float *coord, const int *idx, float *W1, double *T1, double *T2,
                                               void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 0 || k < 1 || c < 1 || g < 1) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < gva_workspace_bytes(n, k, c, g)) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    const int nblk = stage_grid((long long)n * k, TPB * 2);
#define CALL(GG)                                                                                                  \
    hipLaunchKernelGGL(logits_fwd_kernel<GG>, dim3(nblk), dim3(TPB), 0, st, n, k, c, kW, qW, a, b, M, cW, coord, idx, \
                       W1, part)
    {
        // idx, coord, kW (unique rows once), qW in; W1 out
        PtvScopedTimer t(KID_LOGITS_FWD, st, 4.0 * ((double)n * k * (g + 1) + (double)n * (3 + 2 * g)));
        GVA_DISPATCH_G(g, CALL)
    }
#undef CALL
    // part is [nblk][2g]: columns 0..g-1 -> T1, g..2g-1 -> T2 (contiguous in the reduced vector)
    launch_finalize(st, (const float *)part, nblk, 2 * g, MapSplit2<double>{T1, T2, g});
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
