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
#include <cstdlib>

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
// a, b, M are staged in LDS once per workgroup and read back as wave-wide broadcasts (b128): with scalar loads
// straight from memory the deep stages (C = 192 / 384: 18 - 73 KB of M per row) were bound by scalar-cache miss
// latency (104 / 244 us at 72 k / 17 k rows, profiles/r01_*_v8).  SPLIT > 1 spreads the channel loop of one
// 64-row group over the SPLIT waves of the workgroup (partial logits summed through LDS), which gives the small
// deep-stage launches SPLIT x more waves to hide latency with.
#ifndef FWD_RPL
#define FWD_RPL 2
#endif
template <int G, int SPLIT>
__global__ __launch_bounds__(TPB) void logits_fwd_kernel(int n, int k, int c, const float *__restrict__ kW,
                                                         const float *__restrict__ qW, const float *__restrict__ a,
                                                         const float *__restrict__ b, const float *__restrict__ M,
                                                         const float *__restrict__ cW, const float *__restrict__ coord,
                                                         const int *__restrict__ idx, float *__restrict__ W1, float *part,
                                                         unsigned *counter, double *__restrict__ T1,
                                                         double *__restrict__ T2, FoldWFwdArgs F) {
    extern __shared__ float4 lds4[];
    constexpr int G4 = (G + 3) & ~3;
    // rows per lane (SPLIT == 1): the wave-uniform operands of a channel -- (a, b) and the M row, four LDS reads -- serve
    // RPL rows of the lane.  With one row per lane the full-resolution launch was bound by LDS instruction issue (192 broadcast
    // reads per row against 480 vector instructions: `share_active_inst_any` 0.17, `share_wait_inst_any` 0.48 in
    // profiles/r05_final_sq_counters.jsonl, 60 us at 120 k points).  Measured: 60.1 us with one row, 56-57 with two, 58.5
    // with four (142 registers) -- the rest is the row's own chain idx -> coord -> W1 store.
    constexpr int RPL = SPLIT == 1 ? (G <= 8 ? FWD_RPL : 2) : 1;
    constexpr int RPB = TPB / SPLIT * RPL;  // rows per workgroup iteration
    float4 *sAB = lds4;                        // [c]  (a.x, a.y, a.z, b)
    float *sM = (float *)(sAB + c);            // [c][G4]
    float *sRed = sM + (size_t)c * G4;         // [SPLIT][64][G + 1]   (SPLIT > 1 only)
    __shared__ float s_w[WPB][2 * G];
    for (int i = threadIdx.x; i < c; i += TPB) sAB[i] = make_float4(a[3 * i], a[3 * i + 1], a[3 * i + 2], b[i]);
    for (int e = threadIdx.x; e < c * G; e += TPB) sM[(e / G) * G4 + (e % G)] = M[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long rows = (long long)n * k;
    const long long iters = (rows + RPB - 1) / RPB;
    const int cs = (c + SPLIT - 1) / SPLIT;
    const int c0 = SPLIT == 1 ? 0 : wid * cs, c1 = SPLIT == 1 ? c : (c0 + cs < c ? c0 + cs : c);
    const bool finisher = SPLIT == 1 || wid == 0;
    float t1[G], t2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) t1[g] = t2[g] = 0.f;
    for (long long it = blockIdx.x; it < iters; it += gridDim.x) {
        long long row[RPL];
        bool act[RPL];
        int nn[RPL];
        Rel r[RPL];
        float kq[RPL][G], acc[RPL][G];
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            row[j] = it * RPB + (SPLIT == 1 ? j * TPB + (int)threadIdx.x : lane);
            act[j] = row[j] < rows;
            nn[j] = act[j] ? (int)(row[j] / k) : 0;
            r[j].x = r[j].y = r[j].z = 0.f; r[j].src = -1;
            if (act[j]) r[j] = rel_pos(coord, idx, row[j], nn[j]);
        }
        // the neighbour's kW row and the point's qW row are requested before the channel loop, not after it (they need only
        // the neighbour id: one exposed memory round trip less per row)
        if (finisher) {
#pragma unroll
            for (int j = 0; j < RPL; ++j)
#pragma unroll
                for (int g = 0; g < G; ++g)
                    kq[j][g] = ptv2_ld_or_zero(kW + (long long)r[j].src * G + g, act[j] && r[j].src >= 0) -
                               ptv2_ld_or_zero(qW + (long long)nn[j] * G + g, act[j]);
        }
#pragma unroll
        for (int j = 0; j < RPL; ++j)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[j][g] = 0.f;
        for (int ci = c0; ci < c1; ++ci) {
            const float4 ab = sAB[ci];
            float p[RPL];
#pragma unroll
            for (int j = 0; j < RPL; ++j) p[j] = pe_act(ab.x, ab.y, ab.z, ab.w, r[j].x, r[j].y, r[j].z);
            const float *mr = sM + (size_t)ci * G4;
            if (G % 4 == 0) {
#pragma unroll
                for (int g = 0; g < G; g += 4) {
                    const float4 m4 = *(const float4 *)(mr + g);
#pragma unroll
                    for (int j = 0; j < RPL; ++j) {
                        acc[j][g] = __builtin_fmaf(p[j], m4.x, acc[j][g]); acc[j][g + 1] = __builtin_fmaf(p[j], m4.y, acc[j][g + 1]);
                        acc[j][g + 2] = __builtin_fmaf(p[j], m4.z, acc[j][g + 2]); acc[j][g + 3] = __builtin_fmaf(p[j], m4.w, acc[j][g + 3]);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < G; g += 2) {
                    const float2 m2 = *(const float2 *)(mr + g);
#pragma unroll
                    for (int j = 0; j < RPL; ++j) {
                        acc[j][g] = __builtin_fmaf(p[j], m2.x, acc[j][g]); acc[j][g + 1] = __builtin_fmaf(p[j], m2.y, acc[j][g + 1]);
                    }
                }
            }
        }
        if (SPLIT > 1) {
            float *mine = sRed + ((size_t)wid * WAVE + lane) * (G + 1);
#pragma unroll
            for (int g = 0; g < G; ++g) mine[g] = acc[0][g];
            __syncthreads();
            if (wid == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < SPLIT; ++w) t += sRed[((size_t)w * WAVE + lane) * (G + 1) + g];
                    acc[0][g] = t;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            if (finisher && act[j]) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    acc[j][g] += kq[j][g] + cW[g];
                    t1[g] += acc[j][g];
                    t2[g] = __builtin_fmaf(acc[j][g], acc[j][g], t2[g]);
                }
                float *o = W1 + row[j] * G;
                if (G % 4 == 0) {
#pragma unroll
                    for (int g = 0; g < G; g += 4) *(float4 *)(o + g) = make_float4(acc[j][g], acc[j][g + 1], acc[j][g + 2], acc[j][g + 3]);
                } else {
#pragma unroll
                    for (int g = 0; g < G; g += 2) *(float2 *)(o + g) = make_float2(acc[j][g], acc[j][g + 1]);
                }
            }
        }
        if (SPLIT > 1) __syncthreads();  // sRed is rewritten by the next iteration
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float v1 = wave_sum(t1[g]), v2 = wave_sum(t2[g]);
        if (lane == 0) { s_w[wid][g] = v1; s_w[wid][G + g] = v2; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * G) {
        float v = 0.f;
        for (int w = 0; w < WPB; ++w) v += s_w[w][threadIdx.x];
        part_store(part + (size_t)blockIdx.x * 2 * G + threadIdx.x, v);
    }
    // small grids finish their own column sums (counter != NULL); large ones leave them to finalize_kernel
    if (counter && last_block_arrives(counter)) finalize_logit_sums(part, gridDim.x, G, T1, T2, F);
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

// (mu, cov) of the masked relative positions in one finalize: column sums of the records in float64 (9 columns x 113
// record slices, slice sums combined in slice order), then mu = s1 / rows, cov = s2 / rows - mu mu^T as the python path
// computes them (separately rounded product and difference)
__global__ __launch_bounds__(1024) void pos_moments_finalize_kernel(const float *__restrict__ part, int nblk, double rows,
                                                                    double *__restrict__ mu, double *__restrict__ cov) {
    constexpr int SL = 113;
    __shared__ double s_acc[SL][9];
    __shared__ double s_sum[9];
    const int col = threadIdx.x % 9, sl = threadIdx.x / 9;
    if (sl < SL) {
        double a = 0.0;
        for (int b = sl; b < nblk; b += SL) a += (double)part[(size_t)b * 9 + col];
        s_acc[sl][col] = a;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double v = 0.0;
        for (int t = 0; t < SL; ++t) v += s_acc[t][threadIdx.x];
        s_sum[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int r[6] = {0, 0, 0, 1, 1, 2}, c[6] = {0, 1, 2, 1, 2, 2};
        if (threadIdx.x < 3) mu[threadIdx.x] = s_sum[threadIdx.x] / rows;
        if (threadIdx.x >= 3) {
            const int a = r[threadIdx.x - 3], b = c[threadIdx.x - 3];
            const double ma = s_sum[a] / rows, mb = s_sum[b] / rows;
            const double v = __dsub_rn(s_sum[threadIdx.x] / rows, __dmul_rn(ma, mb));
            cov[a * 3 + b] = v;
            cov[b * 3 + a] = v;
        }
    }
}

extern "C" int gva_pos_moments_hip_launcher(int n, int k, const float *coord, const int *idx, double *mu, double *cov,
                                            void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 1 || k < 1 || !coord || !idx || !mu || !cov) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < sizeof(float) * 9 * 2048) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    const int nblk = stage_grid((long long)n * k, TPB * 4);
    hipLaunchKernelGGL(pos_stats_kernel, dim3(nblk), dim3(TPB), 0, st, n, k, coord, idx, part);
    hipLaunchKernelGGL(pos_moments_finalize_kernel, dim3(1), dim3(1024), 0, st, (const float *)part, nblk, (double)n * k, mu, cov);
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

int gva_logits_fwd_mfma_supported(int k, int c, int g);
int gva_logits_fwd_mfma_launch(int n, int k, int c, int g, const float *kW, const float *qW, const float *a, const float *b,
                               const float *M, const float *cW, const float *coord, const int *idx, float *W1, float *part,
                               double *T1, double *T2, const gva::FoldWFwdArgs &F, hipStream_t st);
int gva_logits_point_launch(int n, int k, int c, int g, const float *kW, const float *qW, const float *a, const float *b,
                            const float *M, const float *cW, const float *coord, const int *idx, float *W1, float *part,
                            double *T1, double *T2, const gva::FoldWFwdArgs &F, hipStream_t st);

// F.sc != NULL: the final reduction also folds BN_w (block runtime); the C entry point below passes none
int gva_logits_forward_fold(int n, int k, int c, int g, const float *kW, const float *qW, const float *a, const float *b,
                            const float *M, const float *cW, const float *coord, const int *idx, float *W1, double *T1, double *T2,
                            const gva::FoldWFwdArgs &F, void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 0 || k < 1 || c < 1 || g < 1) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < gva_workspace_bytes(n, k, c, g)) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    const long long rows = (long long)n * k;
    {
        if (gva_logits_fwd_mfma_supported(k, c, g) && !getenv("AO_AMD_BWD_STAGED")) {
            PtvScopedTimer t(KID_LOGITS_FWD, st, 4.0 * ((double)n * k * (g + 1) + (double)n * (3 + 2 * g)));
            const int rc = gva_logits_fwd_mfma_launch(n, k, c, g, kW, qW, a, b, M, cW, coord, idx, W1, part, T1, T2, F, st);
            if (rc != PTV2_OK) return rc;
            PTV2_CHECK_LAUNCH();
            return PTV2_OK;
        }
    }
    if (k <= 16 && c % 4 == 0 && (g == 48 || g == 64) && !getenv("AO_AMD_BWD_STAGED")) {  // pays for wide G only
        PtvScopedTimer t(KID_LOGITS_FWD, st, 4.0 * ((double)n * k * (g + 1) + (double)n * (3 + 2 * g)));
        const int rc = gva_logits_point_launch(n, k, c, g, kW, qW, a, b, M, cW, coord, idx, W1, part, T1, T2, F, st);
        if (rc != PTV2_OK) return rc;
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    const int g4 = (g + 3) & ~3;
    const size_t lds_base = sizeof(float4) * (size_t)c + sizeof(float) * (size_t)c * g4;
    const size_t lds_red = sizeof(float) * 4 * WAVE * (g + 1);
    // small launches: 4 waves per 64 rows instead of 1 (when the cross-wave reduction buffer still fits in LDS)
    const bool split = rows < 400000 && lds_base + lds_red <= 96 * 1024;
    const size_t lds = lds_base + (split ? lds_red : 0);
    if (lds > 150 * 1024 || g % 2 != 0) return PTV2_ERR_ARG;
    const int nblk = stage_grid(rows, split ? WAVE : (g <= 8 ? FWD_RPL : 2) * TPB);  // (the unsplit kernel takes several rows per lane)
    const bool own_final = (size_t)nblk * 2 * g <= FUSED_FINAL_MAX;
    unsigned *cnt = own_final ? ptv2_stream_counters(st) : nullptr;
    if (own_final && !cnt) return PTV2_ERR_LAUNCH;
#define CALL(GG)                                                                                                   \
    if (split) {                                                                                                   \
        if (lds > 32 * 1024)                                                                                       \
            (void)hipFuncSetAttribute((const void *)logits_fwd_kernel<GG, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                   \
        hipLaunchKernelGGL((logits_fwd_kernel<GG, 4>), dim3(nblk), dim3(TPB), lds, st, n, k, c, kW, qW, a, b, M, cW, coord, \
                           idx, W1, part, cnt ? cnt + CNT_LOGITS_FWD : nullptr, T1, T2, F);                                                                         \
    } else {                                                                                                       \
        if (lds > 32 * 1024)                                                                                       \
            (void)hipFuncSetAttribute((const void *)logits_fwd_kernel<GG, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                   \
        hipLaunchKernelGGL((logits_fwd_kernel<GG, 1>), dim3(nblk), dim3(TPB), lds, st, n, k, c, kW, qW, a, b, M, cW, coord, \
                           idx, W1, part, cnt ? cnt + CNT_LOGITS_FWD : nullptr, T1, T2, F);                                                                         \
    }
    {
        // idx, coord, kW (unique rows once), qW in; W1 out
        PtvScopedTimer t(KID_LOGITS_FWD, st, 4.0 * ((double)n * k * (g + 1) + (double)n * (3 + 2 * g)));
        GVA_DISPATCH_G(g, CALL)
    }
#undef CALL
    if (!own_final) hipLaunchKernelGGL(finalize_logit_sums_kernel, dim3((g + FLS_GROUPS - 1) / FLS_GROUPS), dim3(1024), 0, st, (const float *)part, nblk, g, T1, T2, F);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int gva_logits_forward_hip_launcher(int n, int k, int c, int g, const float *kW, const float *qW,
                                               const float *a, const float *b, const float *M, const float *cW,
                                               const float *coord, const int *idx, float *W1, double *T1, double *T2,
                                               void *workspace, size_t workspace_bytes, void *stream) {
    return gva_logits_forward_fold(n, k, c, g, kW, qW, a, b, M, cW, coord, idx, W1, T1, T2, gva::FoldWFwdArgs{}, workspace,
                                   workspace_bytes, stream);
}
