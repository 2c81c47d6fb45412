// ao_amd/csrc/gva_bwd.hip -- backward stages of the fused grouped vector attention (gfx950).
// Math: ao_amd/ptv2/gva.py; forward kernels: gva_fwd.hip.
//
// Scatter-adds into per-point rows (grad of kW, grad of v) are NOT done with float atomics: the caller
// passes the inverse neighbour table (for each point j the slots (n,s) with idx[n,s] == j, CSR,
// ascending slot order) and a gather kernel sums each destination row in that fixed order --
// bitwise reproducible, and ~5x cheaper than 1.3 TB/s atomics at these row sizes.  Parameter
// gradients are per-wave / per-block partial sums reduced in fixed order by a final kernel.
//
//   logits backward   rows kernel   gWt = gW1 + gT1 + 2 W1 gT2 (BN_w statistics path), grad cW
//                     gather kernel grad kW (via inverse table), grad qW
//                     params kernel grad M, grad a, grad b   (lanes over channels, rows streamed
//                                   through LDS as broadcast operands: no cross-lane reductions)
//   (aggregate backward: gva_aggregate.hip)
#include <algorithm>

#include <cstdlib>

#include "gva_common.h"

namespace gva {

struct AggLds {  // same helpers as gva_fwd.hip
    __host__ __device__ static constexpr int gp(int G) { return (G | 1) + ((G & 1) ? 2 : 0); }
    __host__ __device__ static constexpr int G4(int G) { return (G + 3) & ~3; }
    __host__ __device__ static constexpr size_t r4(size_t v) { return (v + 3) & ~(size_t)3; }
};

inline int stage_grid(long long work_items, int per_block, int cap = MAX_BLOCKS) {
    long long b = (work_items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
inline bool pow2(int k) { return k > 0 && (k & (k - 1)) == 0; }

// ============================================================ logits backward ==
// gWt = gW1 + gT1 + 2 gT2 W1 (the two BN_w statistics paths folded into the row gradient) and its column sums.
// Flat float4 streaming: a thread owns whole "units" of lcm(4, G) consecutive floats, so the column of every
// element of its float4s is a compile-time constant and consecutive lanes touch consecutive memory (the previous
// row-per-lane form read 4-byte pieces 4 G bytes apart and ran at 1 TB/s).
template <int G>
__global__ __launch_bounds__(TPB) void logits_bwd_rows_kernel(long long rows, const float *__restrict__ W1,
                                                              const float *__restrict__ gW1,
                                                              const double *__restrict__ gT1,
                                                              const double *__restrict__ gT2, float *__restrict__ gWt,
                                                              float *part, unsigned *counter, float *__restrict__ gcW,
                                                              FoldWBwdArgs F) {
    constexpr int U = (G % 4 == 0) ? G : (G % 2 == 0 ? 2 * G : 4 * G);  // floats per unit = lcm(4, G)
    constexpr int UQ = U / 4;                                           // float4 per unit
    __shared__ float s_w[WPB][G];
    float t[G], c1[G], c2[G];
    if (F.gsc) {  // gT1 / gT2 from the BatchNorm fold's backward, evaluated here: one group per thread, shared through LDS
        __shared__ float s_c[2][G];
        if (threadIdx.x < G) {
            double t1, t2;
            float gg, gb;
            fold_w_bwd_channel(F, threadIdx.x, t1, t2, gg, gb);
            s_c[0][threadIdx.x] = (float)t1;
            s_c[1][threadIdx.x] = 2.f * (float)t2;
            if (blockIdx.x == 0) { F.ggamma[threadIdx.x] = gg; F.gbeta[threadIdx.x] = gb; }
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; ++g) { t[g] = 0.f; c1[g] = s_c[0][g]; c2[g] = s_c[1][g]; }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) { t[g] = 0.f; c1[g] = (float)gT1[g]; c2[g] = 2.f * (float)gT2[g]; }
    }
    const long long total = rows * G, units = total / U;  // rows * G is a multiple of U whenever rows % (U / G) == 0
    for (long long u = (long long)blockIdx.x * TPB + threadIdx.x; u < units; u += (long long)gridDim.x * TPB) {
        const float4 *pw = (const float4 *)(W1 + u * U), *pg = (const float4 *)(gW1 + u * U);
        float4 *po = (float4 *)(gWt + u * U);
#pragma unroll
        for (int j = 0; j < UQ; ++j) {
            const float4 w4 = pw[j], g4 = pg[j];
            float4 o;
            o.x = __builtin_fmaf(w4.x, c2[(4 * j) % G], g4.x + c1[(4 * j) % G]);
            o.y = __builtin_fmaf(w4.y, c2[(4 * j + 1) % G], g4.y + c1[(4 * j + 1) % G]);
            o.z = __builtin_fmaf(w4.z, c2[(4 * j + 2) % G], g4.z + c1[(4 * j + 2) % G]);
            o.w = __builtin_fmaf(w4.w, c2[(4 * j + 3) % G], g4.w + c1[(4 * j + 3) % G]);
            po[j] = o;
            t[(4 * j) % G] += o.x; t[(4 * j + 1) % G] += o.y; t[(4 * j + 2) % G] += o.z; t[(4 * j + 3) % G] += o.w;
        }
    }
    // tail rows (rows not a multiple of U / G): element-wise by the first threads of block 0
    if (blockIdx.x == 0) {
        for (long long e = units * U + threadIdx.x; e < total; e += TPB) {
            const int g = (int)(e % G);
            const float v = __builtin_fmaf(W1[e], c2[g], gW1[e] + c1[g]);
            gWt[e] = v;
#pragma unroll
            for (int gg = 0; gg < G; ++gg) t[gg] += gg == g ? v : 0.f;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float v = wave_sum(t[g]);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6][g] = v;
    }
    __syncthreads();
    if (threadIdx.x < G) {
        float v = 0.f;
        for (int w = 0; w < WPB; ++w) v += s_w[w][threadIdx.x];
        part_store(part + (size_t)blockIdx.x * G + threadIdx.x, v);
    }
    if (counter && last_block_arrives(counter)) finalize_columns(part, gridDim.x, G, MapVec<float>{gcW});
}

// grad qW[j,g] = -sum_s gWt[j,s,g];  grad kW[j,g] = sum over slots that point at j
__global__ __launch_bounds__(TPB) void logits_bwd_gather_kernel(int n, int k, int g, const float *__restrict__ gWt,
                                                                const int *__restrict__ idx,
                                                                const int *__restrict__ inv_ptr,
                                                                const int *__restrict__ inv_rows,
                                                                float *__restrict__ gkW, float *__restrict__ gqW) {
    const long long total = (long long)n * g;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const int j = (int)(e / g), gi = (int)(e - (long long)j * g);
        float q = 0.f;
        if (k == 16) {  // all 16 slot loads in flight before the first add
            float t[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) t[s] = gWt[((long long)j * 16 + s) * g + gi];
#pragma unroll
            for (int s = 0; s < 16; ++s) q += t[s];
        } else {
            for (int s = 0; s < k; ++s) q += gWt[((long long)j * k + s) * g + gi];
        }
        gqW[e] = -q;
        if (inv_ptr) {
            float acc = 0.f;
            const int p0 = inv_ptr[j], p1 = inv_ptr[j + 1];
            constexpr int UB = 8;  // 8 list entries at a time: ids, then values, then the sums in list order
            // (unconditional loads -- clamped index with -1 or'ed in, zero pad: common.h ptv2_zero_pad -- so that a batch is one
            // round trip, not eight)
            int r[UB], rn[UB];
            const int plast = p1 > 0 ? p1 - 1 : 0;
#pragma unroll
            for (int u = 0; u < UB; ++u) r[u] = inv_rows[p0 + u < p1 ? p0 + u : plast] | (p0 + u < p1 ? 0 : -1);
            for (int p = p0; p < p1; p += UB) {
                float t[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) rn[u] = inv_rows[p + UB + u < p1 ? p + UB + u : plast] | (p + UB + u < p1 ? 0 : -1);  // next batch's ids ride along
#pragma unroll
                for (int u = 0; u < UB; ++u) t[u] = ptv2_ld_or_zero(gWt + (long long)r[u] * g + gi, r[u] >= 0);
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (r[u] >= 0) acc += t[u];
#pragma unroll
                for (int u = 0; u < UB; ++u) r[u] = rn[u];
            }
            gkW[e] = acc;
        }
    }
    if (!inv_ptr) {  // fallback without the inverse table: float atomics (sum order unspecified)
        const long long rows_g = (long long)n * k * g;
        for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < rows_g; e += (long long)gridDim.x * TPB) {
            const long long row = e / g;
            const int gi = (int)(e - row * g), src = idx[row];
            if (src >= 0) atomicAdd(gkW + (long long)src * g + gi, gWt[e]);
        }
    }
}

// grad M (C,G), grad a (C,3), grad b (C): thread <-> channel, rows broadcast from LDS.
constexpr int PR_TILE = 128;  // rows per LDS tile

template <int G>
__global__ __launch_bounds__(TPB) void logits_bwd_params_kernel(int n, int k, int c, const float *__restrict__ a,
                                                                const float *__restrict__ b,
                                                                const float *__restrict__ M,
                                                                const float *__restrict__ coord,
                                                                const int *__restrict__ idx,
                                                                const float *__restrict__ gWt,
                                                                float *__restrict__ part) {
    constexpr int G4 = AggLds::G4(G);
    __shared__ float4 sPos[PR_TILE];
    __shared__ __attribute__((aligned(16))) float sG[PR_TILE][G4];
    extern __shared__ float4 dyn4[];  // slice combine buffer: [nsl][cb][G+4] floats
    float *sComb = (float *)dyn4;
    const long long rows = (long long)n * k;
    const int cb = c < TPB ? c : TPB;               // channels per pass
    const int nsl = c < TPB ? TPB / c : 1;          // row slices per pass
    const int slice = threadIdx.x / cb, chl = threadIdx.x - slice * cb;
    const bool worker = slice < nsl;
    const int per = G + 4;
    float *mypart = part + (size_t)blockIdx.x * c * per;
    for (int cbase = 0; cbase < c; cbase += cb) {
        const int ch = cbase + chl;
        const bool act = worker && ch < c;
        float Mr[G], gMr[G], ga0 = 0.f, ga1 = 0.f, ga2 = 0.f, gb0 = 0.f;
        float ax = 0.f, ay = 0.f, az = 0.f, bb = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) { Mr[g] = act ? M[ch * G + g] : 0.f; gMr[g] = 0.f; }
        if (act) { ax = a[3 * ch]; ay = a[3 * ch + 1]; az = a[3 * ch + 2]; bb = b[ch]; }
        for (long long t0 = (long long)blockIdx.x * PR_TILE; t0 < rows; t0 += (long long)gridDim.x * PR_TILE) {
            __syncthreads();
            for (int r = threadIdx.x; r < PR_TILE; r += TPB) {
                long long row = t0 + r;
                float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < rows) {
                    Rel rr = rel_pos(coord, idx, row, (int)(row / k));
                    p = make_float4(rr.x, rr.y, rr.z, 1.f);
                }
                sPos[r] = p;
            }
            for (int e = threadIdx.x; e < PR_TILE * G; e += TPB) {
                int r = e / G, g = e - r * G;
                long long row = t0 + r;
                sG[r][g] = row < rows ? gWt[row * G + g] : 0.f;
            }
            __syncthreads();
            if (act) {
                const int rend = (int)((rows - t0) < PR_TILE ? (rows - t0) : PR_TILE);
                for (int r = slice; r < rend; r += nsl) {
                    const float4 p = sPos[r];
                    const float P = pe_act(ax, ay, az, bb, p.x, p.y, p.z);
                    float dot = 0.f;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float gw = sG[r][g];
                        dot = __builtin_fmaf(gw, Mr[g], dot);
                        gMr[g] = __builtin_fmaf(P, gw, gMr[g]);
                    }
                    const float gpre = P > 0.f ? dot : 0.f;
                    ga0 = __builtin_fmaf(gpre, p.x, ga0);
                    ga1 = __builtin_fmaf(gpre, p.y, ga1);
                    ga2 = __builtin_fmaf(gpre, p.z, ga2);
                    gb0 += gpre;
                }
            }
        }
        // combine the row slices in fixed order, then write this block's partial for these channels
        __syncthreads();
        if (act) {
            float *dst = sComb + ((size_t)slice * cb + chl) * per;
#pragma unroll
            for (int g = 0; g < G; ++g) dst[g] = gMr[g];
            dst[G] = ga0; dst[G + 1] = ga1; dst[G + 2] = ga2; dst[G + 3] = gb0;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < cb * per; e += TPB) {
            const int cl = e / per;
            if (cbase + cl < c) {
                float v = 0.f;
                for (int sl = 0; sl < nsl; ++sl) v += sComb[(size_t)sl * cb * per + e];
                mypart[(size_t)(cbase + cl) * per + (e - cl * per)] = v;
            }
        }
    }
}



}  // namespace gva

using namespace gva;

int gva_bwd_point_supported(int k, int c, int g);
int gva_logits_bwd_fused_supported(int k, int c, int g);
int gva_logits_bwd_fused_launch(int n, int k, int c, int g, const float *a, const float *b, const float *M, const float *coord,
                                const int *idx, const float *W1, const float *gW1, const double *gT1, const double *gT2,
                                const gva::FoldWBwdArgs &F, float *gWt, float *part, size_t part_floats_avail, float *gM, float *ga,
                                float *gb, float *gcW, hipStream_t st);
int gva_logits_params_point_launch(int n, int k, int c, int g, const float *a, const float *b, const float *M,
                                   const float *coord, const int *idx, const float *gWt, float *part, int max_blocks,
                                   int *nblk_out, hipStream_t st);

#define GVA_DISPATCH_G(g, CALL)            \
    switch (g) {                           \
        case 6: { CALL(6); break; }        \
        case 12: { CALL(12); break; }      \
        case 24: { CALL(24); break; }      \
        case 48: { CALL(48); break; }      \
        case 64: { CALL(64); break; }      \
        default: return PTV2_ERR_ARG;      \
    }

extern "C" size_t gva_workspace_bytes(int n, int k, int c, int g);

// F.gsc != NULL: gT1 / gT2 are not read; the rows kernel derives them from the fold_w backward (and writes the BatchNorm's
// parameter gradients) -- internal to the block runtime (gva_block.hip)
int gva_logits_backward_foldw(int n, int k, int c, int g, const float *a, const float *b, const float *M, const float *coord,
                              const int *idx, const float *W1, const float *gW1, const double *gT1, const double *gT2,
                              const gva::FoldWBwdArgs &F, const int *inv_ptr, const int *inv_rows, float *gkW, float *gqW, float *ga,
                              float *gb, float *gM, float *gcW, void *workspace, size_t workspace_bytes, void *stream);

extern "C" int gva_logits_backward_hip_launcher(int n, int k, int c, int g, const float *a, const float *b,
                                                const float *M, const float *coord, const int *idx,
                                                const float *W1, const float *gW1, const double *gT1,
                                                const double *gT2, const int *inv_ptr, const int *inv_rows,
                                                float *gkW, float *gqW, float *ga, float *gb, float *gM, float *gcW,
                                                void *workspace, size_t workspace_bytes, void *stream) {
    return gva_logits_backward_foldw(n, k, c, g, a, b, M, coord, idx, W1, gW1, gT1, gT2, gva::FoldWBwdArgs{}, inv_ptr, inv_rows, gkW,
                                     gqW, ga, gb, gM, gcW, workspace, workspace_bytes, stream);
}

int gva_logits_backward_foldw(int n, int k, int c, int g, const float *a, const float *b, const float *M, const float *coord,
                              const int *idx, const float *W1, const float *gW1, const double *gT1, const double *gT2,
                              const gva::FoldWBwdArgs &F, const int *inv_ptr, const int *inv_rows, float *gkW, float *gqW, float *ga,
                              float *gb, float *gM, float *gcW, void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 0 || k < 1 || c < 1 || g < 1) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < gva_workspace_bytes(n, k, c, g)) return PTV2_ERR_WORKSPACE;
    if (n == 0) return PTV2_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)n * k;
    float *part = (float *)workspace;
    float *gWt = (float *)((char *)workspace + rows_offset_bytes(c, g));
    // wide-group levels: rows + parameter gradients in one pipelined MFMA launch (gva_bwd_logits.hip), then the gather
    const char *lb = getenv("AO_AMD_LOGITS_BWD");  // "staged": the three-kernel form (A/B switch of the tests)
    const bool fused_off = lb && lb[0] == 's';
    if (inv_ptr && gva_logits_bwd_fused_supported(k, c, g) && !fused_off && !getenv("AO_AMD_BWD_STAGED")) {
        {
            // W1, gW1 in, gWt out, idx, coord; parameter-sized outputs
            PtvScopedTimer t(KID_LOGITS_BWD_FUSED + (g == 6 ? 0 : g == 12 ? 1 : g == 24 ? 2 : g == 48 ? 3 : 4), st,
                             4.0 * ((double)rows * (3 * g + 1) + 3.0 * n));
            const int rc = gva_logits_bwd_fused_launch(n, k, c, g, a, b, M, coord, idx, W1, gW1, gT1, gT2, F, gWt, part,
                                                       part_floats(c, g), gM, ga, gb, gcW, st);
            if (rc != PTV2_OK) return rc;
        }
        {
            PtvScopedTimer t(KID_LOGITS_BWD_GATHER, st, 4.0 * ((double)rows * g + 2.0 * n * g + rows));
            hipLaunchKernelGGL(logits_bwd_gather_kernel, dim3(stage_grid((long long)n * g, TPB)), dim3(TPB), 0, st, n, k, g,
                               (const float *)gWt, idx, inv_ptr, inv_rows, gkW, gqW);
        }
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    // ~16 float4 per thread: with 4 the launch was thousands of single-shot workgroups whose records the last one to
    // arrive then had to sum (83-88 % of the wave cycles parked; 13.45 -> 13.35 ms per step with a quarter of the grid)
    const int nb_rows = stage_grid(rows * g / 64, TPB);
#define CALL(GG) \
    hipLaunchKernelGGL(logits_bwd_rows_kernel<GG>, dim3(nb_rows), dim3(TPB), 0, st, rows, W1, gW1, gT1, gT2, gWt, part, \
                       cnt ? cnt + CNT_LOGITS_BWD_ROWS : nullptr, gcW, F)
    const bool own_final = (size_t)nb_rows * g <= FUSED_FINAL_MAX;
    unsigned *cnt = own_final ? ptv2_stream_counters(st) : nullptr;
    if (own_final && !cnt) return PTV2_ERR_LAUNCH;
    GVA_DISPATCH_G(g, CALL)
#undef CALL
    if (!own_final) launch_finalize(st, (const float *)part, nb_rows, g, MapVec<float>{gcW});
    hipLaunchKernelGGL(logits_bwd_gather_kernel, dim3(stage_grid((long long)n * g, TPB)), dim3(TPB), 0, st, n, k, g,
                       (const float *)gWt, idx, inv_ptr, inv_rows, gkW, gqW);
    // params kernel: its partials go after the rows-kernel partials (still inside the partial region)
    float *ppart = part + (size_t)nb_rows * g;
    const int cbk = c < TPB ? c : TPB, nsl = c < TPB ? TPB / c : 1;
    const size_t comb_bytes = sizeof(float) * (size_t)nsl * cbk * (g + 4);
    // enough workgroups to hide the tile-load latency (8 per CU), bounded by the partial-sum budget
    const int par_cap = (int)std::max<long long>(64, std::min<long long>(MAX_BLOCKS, ((long long)MAX_PARAM_BLOCKS * 24576) / ((long long)c * (g + 4))));
    if (g >= 48 && gva_bwd_point_supported(k, c, g) && !getenv("AO_AMD_BWD_STAGED")) {  // pays for wide G only
        int nb = 0;
        {
            PtvScopedTimer t(KID_LOGITS_BWD_PARAMS, st, 4.0 * ((double)rows * (g + 1) + 3.0 * n));
            const int rc = gva_logits_params_point_launch(n, k, c, g, a, b, M, coord, idx, gWt, ppart, par_cap, &nb, st);
            if (rc != PTV2_OK) return rc;
        }
        launch_finalize(st, (const float *)ppart, nb, c * (g + 4), MapLogitsParams{gM, ga, gb, g});
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    const int nb_par = stage_grid(rows, PR_TILE, par_cap);
#define CALL(GG)                                                                                                   \
    if (comb_bytes > 32 * 1024)                                                                                    \
        (void)hipFuncSetAttribute((const void *)logits_bwd_params_kernel<GG>,                                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)comb_bytes);                    \
    hipLaunchKernelGGL(logits_bwd_params_kernel<GG>, dim3(nb_par), dim3(TPB), comb_bytes, st, n, k, c, a, b, M, coord, idx, \
                       (const float *)gWt, ppart)
    {
        const int passes = (c + TPB - 1) / TPB;  // the row stream is re-read once per 256-channel pass
        PtvScopedTimer t(KID_LOGITS_BWD_PARAMS, st, 4.0 * passes * ((double)rows * (g + 1) + 3.0 * n));
        GVA_DISPATCH_G(g, CALL)
    }
#undef CALL
    launch_finalize(st, (const float *)ppart, nb_par, c * (g + 4), MapLogitsParams{gM, ga, gb, g});
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
