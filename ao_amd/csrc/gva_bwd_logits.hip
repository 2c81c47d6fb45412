// ao_amd/csrc/gva_bwd_logits.hip -- logits backward of grouped vector attention for the wide-group levels (G >= 12):
// the BatchNorm-fold of the row gradient AND the parameter gradients of the logits stage in ONE streaming launch on the
// matrix cores.
//
// Round 2 ran this stage as three kernels per attention block (gva_bwd.hip):
//   logits_bwd_rows    gWt = gW1 + gT1 + 2 gT2 W1, column sums -> grad cW                      (N K G) in, (N K G) out
//   logits_bwd_gather  grad kW (inverse table), grad qW                                         reads gWt
//   logits_bwd_params  grad M (C,G) = P^T gWt, (grad a, grad b) = relu'(P) (gWt M^T) (pos, 1)   reads gWt again
// with the third one on the vector ALU (thread <-> channel, two G-long register rows: 664 M FMAs at 4.5 k points = 17 us at
// 100 % VALU utilisation, 50 us measured) and a point-per-wavefront MFMA form that tied with it because every point paid
// its load latencies in sequence (idx -> coord -> gWt in two layouts) behind two workgroup barriers.
//
// Here a wavefront keeps the K = 16 slots of a point as one MFMA dimension (as gva_bwd_point.hip) and the loop is software
// pipelined: while point i is on the matrix cores the W1 / gW1 rows of point i+1 (both operand layouts), its neighbour
// coordinates and the neighbour ids of point i+2 are in flight; positions go through a wave-private LDS record (no
// workgroup barrier anywhere in the loop).  The row gradient gWt is formed in registers from W1 and gW1 (the same fused
// multiply-add as the rows kernel: bit-identical values), written once for the gather kernel, and consumed in place:
//   D (s,ch)   = gWt (s,g) M^T (g,ch)          -> gpre = relu'(P) D -> (ga, gb)[ch] += gpre (pos, 1)
//   gM (ch,g) += P^T (ch,s) gWt (s,g)                                 accumulated in registers over the workgroup's points
//   gcW (g)   += sum_s gWt (s,g)
// NW wavefronts share a point, each owning C / NW channels (results of a product feed the next without leaving registers:
// the D tile of the first product is in the layout the mask and the (ga, gb) sums want).
#include <algorithm>

#include "gva_common.h"

namespace gva {

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mfma4l(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int G, int C, int NW>
struct LogitsBwdCfg {
    static constexpr int GT = (G + 15) / 16, PW = 4 / NW, CW = C / NW, UT = CW / 16, PER = G + 4;
    static constexpr bool HOIST_M = UT * GT <= 12;  // M fragments of this wavefront's channels stay in registers
    static constexpr size_t REC = (size_t)C * PER + GT * 16;  // floats of a workgroup record: [C][G+4], then gcW (padded)
};

// part[blockIdx.x][REC]
template <int G, int C, int NW>
__global__ __launch_bounds__(256) void logits_bwd_fused_kernel(int n, int k, const float *__restrict__ a,
                                                               const float *__restrict__ b, const float *__restrict__ M,
                                                               const float *__restrict__ coord, const int *__restrict__ idx,
                                                               const float *__restrict__ W1, const float *__restrict__ gW1,
                                                               const double *__restrict__ gT1, const double *__restrict__ gT2,
                                                               float *__restrict__ gWt, float *__restrict__ part, FoldWBwdArgs F) {
    using K = LogitsBwdCfg<G, C, NW>;
    constexpr int GT = K::GT, PW = K::PW, CW = K::CW, UT = K::UT, PER = K::PER, G16 = GT * 16;
    static_assert(G % 4 == 0 && CW % 16 == 0, "layout A loads float4 along g; a wavefront owns whole 16-channel tiles");
    extern __shared__ float4 lds4[];
    float4 *sAB = lds4;                                  // [C] (a.xyz, b)
    float4 *sPos = sAB + C;                              // [4 waves][2][16] wave-private position records
    float *sC1 = (float *)(sPos + 4 * 2 * 16);           // [G16]
    float *sC2 = sC1 + G16;                              // [G16]
    float *sFin = sC2 + G16;                             // [REC]   (PW > 1 only: cross-point-slot sums)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = wid / NW, sub = wid % NW;
    const int l15 = lane & 15, q = lane >> 4;
    const int c0 = sub * CW;

    for (int ch = tid; ch < C; ch += 256) sAB[ch] = make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]);
    if (tid < G16) {  // c1 = gT1, c2 = 2 gT2: the two statistics paths of BN_w folded into every row gradient
        float c1 = 0.f, c2 = 0.f;
        if (tid < G) {
            if (F.gsc) {
                double t1, t2;
                float gg, gb_;
                fold_w_bwd_channel(F, tid, t1, t2, gg, gb_);
                c1 = (float)t1;
                c2 = 2.f * (float)t2;
                if (blockIdx.x == 0) { F.ggamma[tid] = gg; F.gbeta[tid] = gb_; }
            } else {
                c1 = (float)gT1[tid];
                c2 = 2.f * (float)gT2[tid];
            }
        }
        sC1[tid] = c1;
        sC2[tid] = c2;
    }
    if (PW > 1)
        for (int e = tid; e < (int)K::REC; e += 256) sFin[e] = 0.f;
    __syncthreads();

    // constants of this lane: folding constants in both layouts, (a, b) of its channels, M fragments
    float4 c1A[GT], c2A[GT];
    float c1B[GT], c2B[GT];
#pragma unroll
    for (int t = 0; t < GT; ++t) {
        c1A[t] = *(const float4 *)(sC1 + 16 * t + 4 * q);
        c2A[t] = *(const float4 *)(sC2 + 16 * t + 4 * q);
        c1B[t] = sC1[16 * t + l15];
        c2B[t] = sC2[16 * t + l15];
    }
    float4 mreg[K::HOIST_M ? UT : 1][K::HOIST_M ? GT : 1];
    if (K::HOIST_M) {
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const int g0 = 16 * t + 4 * q;
                mreg[u][t] = g0 < G ? *(const float4 *)(M + (size_t)(c0 + 16 * u + l15) * G + g0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }

    float4 accAB[UT];
    v4f accM[UT][GT];
    float tcw[GT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        accAB[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < GT; ++t) accM[u][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t = 0; t < GT; ++t) tcw[t] = 0.f;

    // ---- loads of one point: rows of W1 / gW1 in the two operand layouts
    // (every load of the loop is unconditional -- the zero pad of common.h for masked lanes, all four lane quarters carrying
    // slot l15's id and position: see logits_bwd_fused6_kernel below)
    struct Rows { float4 wA[GT], gA[GT]; float wB[4][GT], gB[4][GT]; };
    auto load_rows = [&](long long pt, Rows &R) {
        const bool act = pt < n;
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int g0 = 16 * t + 4 * q;
            const bool okA = act && l15 < k && g0 < G;
            const size_t oA = ((size_t)pt * k + l15) * G + g0;
            R.wA[t] = ptv2_ld_or_zero((const float4 *)(W1 + oA), okA);
            R.gA[t] = ptv2_ld_or_zero((const float4 *)(gW1 + oA), okA);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int s = 4 * st + q, g = 16 * t + l15;
                const bool okB = act && s < k && g < G;
                const size_t o = ((size_t)pt * k + s) * G + g;
                R.wB[st][t] = ptv2_ld_or_zero(W1 + o, okB);
                R.gB[st][t] = ptv2_ld_or_zero(gW1 + o, okB);
            }
        }
    };
    const long long stride = (long long)gridDim.x * PW;
    float4 *myPos = sPos + wid * 32;
    const long long lastp = (long long)n - 1;
    const int lk = l15 < k ? l15 : 0;
    // neighbour ids two points ahead, coordinates one point ahead (every lane quarter: its own copy, no barrier)
    auto load_idx = [&](long long pt) -> int { return idx[(pt < n ? pt : lastp) * k + lk]; };
    struct Raw { float sx, sy, sz, px, py, pz; bool ok; };
    auto load_raw = [&](long long pt, int src) -> Raw {
        const long long ps = src >= 0 ? src : 0, pp = pt < n ? pt : lastp;
        Raw r;
        r.sx = coord[3 * ps]; r.sy = coord[3 * ps + 1]; r.sz = coord[3 * ps + 2];
        r.px = coord[3 * pp]; r.py = coord[3 * pp + 1]; r.pz = coord[3 * pp + 2];
        r.ok = pt < n && l15 < k && src >= 0;
        return r;
    };
    auto rel_of = [](const Raw &r) { return r.ok ? make_float4(r.sx - r.px, r.sy - r.py, r.sz - r.pz, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f); };
    const long long pt0 = (long long)blockIdx.x * PW + p;
    Rows Rn;
    load_rows(pt0, Rn);
    int idx_n = load_idx(pt0 + stride);
    myPos[l15] = rel_of(load_raw(pt0, load_idx(pt0)));
    int cur = 0;
    for (long long pt = pt0; pt < n; pt += stride, cur ^= 1) {  // (no workgroup barrier inside: trip counts may differ per wavefront)
        const bool act = pt < n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();  // record `cur` of this wavefront is in LDS
        // ---- current point: gWt in both layouts from the prefetched rows
        float uA[GT][4], uB[4][GT];
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            uA[t][0] = __builtin_fmaf(Rn.wA[t].x, c2A[t].x, Rn.gA[t].x + c1A[t].x);
            uA[t][1] = __builtin_fmaf(Rn.wA[t].y, c2A[t].y, Rn.gA[t].y + c1A[t].y);
            uA[t][2] = __builtin_fmaf(Rn.wA[t].z, c2A[t].z, Rn.gA[t].z + c1A[t].z);
            uA[t][3] = __builtin_fmaf(Rn.wA[t].w, c2A[t].w, Rn.gA[t].w + c1A[t].w);
            const int g0 = 16 * t + 4 * q;
            const bool ok = act && l15 < k && g0 < G;
            if (!ok) uA[t][0] = uA[t][1] = uA[t][2] = uA[t][3] = 0.f;
            if (ok && sub == 0) *(float4 *)(gWt + ((size_t)pt * k + l15) * G + g0) = make_float4(uA[t][0], uA[t][1], uA[t][2], uA[t][3]);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bool okb = act && 4 * st + q < k && 16 * t + l15 < G;
                uB[st][t] = okb ? __builtin_fmaf(Rn.wB[st][t], c2B[t], Rn.gB[st][t] + c1B[t]) : 0.f;
                if (sub == 0) tcw[t] += uB[st][t];
            }
        }
        // ---- requests for the next points
        load_rows(pt + stride, Rn);
        const Raw raw_n = load_raw(pt + stride, idx_n);
        idx_n = load_idx(pt + 2 * stride);
        // ---- positions of this point in the two row layouts
        float4 rp[4], pq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { rp[r] = myPos[cur * 16 + 4 * q + r]; pq[r] = myPos[cur * 16 + 4 * r + q]; }
        float4 mnext[GT];  // (not hoisted: the M fragments of channel tile u+1 are requested while tile u computes)
        auto load_m = [&](int u, float4 (&m)[GT]) {
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const int g0 = 16 * t + 4 * q;
                m[t] = ptv2_ld_or_zero((const float4 *)(M + (size_t)(c0 + 16 * u + l15) * G + g0), g0 < G);
            }
        };
        if (!K::HOIST_M) load_m(0, mnext);
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const int ch = c0 + 16 * u + l15;
            const float4 ab = sAB[ch];
            v4f d = (v4f){0.f, 0.f, 0.f, 0.f};
            float4 mcur[GT];
            if (!K::HOIST_M) {
#pragma unroll
                for (int t = 0; t < GT; ++t) mcur[t] = mnext[t];
                if (u + 1 < UT) load_m(u + 1, mnext);
            }
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const float4 m4 = K::HOIST_M ? mreg[u][t] : mcur[t];
                d = mfma4l(uA[t][0], m4.x, d);
                d = mfma4l(uA[t][1], m4.y, d);
                d = mfma4l(uA[t][2], m4.z, d);
                d = mfma4l(uA[t][3], m4.w, d);
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {  // gM (ch, g) += P^T gWt: contraction over the slots s = 4 st + q
                const float P = pe_act(ab.x, ab.y, ab.z, ab.w, pq[st].x, pq[st].y, pq[st].z);
#pragma unroll
                for (int t = 0; t < GT; ++t) accM[u][t] = mfma4l(P, uB[st][t], accM[u][t]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // rows s = 4 q + r of column ch
                const float P = pe_act(ab.x, ab.y, ab.z, ab.w, rp[r].x, rp[r].y, rp[r].z);
                const float gpre = (P > 0.f && act && 4 * q + r < k) ? d[r] : 0.f;
                accAB[u].x = __builtin_fmaf(gpre, rp[r].x, accAB[u].x);
                accAB[u].y = __builtin_fmaf(gpre, rp[r].y, accAB[u].y);
                accAB[u].z = __builtin_fmaf(gpre, rp[r].z, accAB[u].z);
                accAB[u].w += gpre;
            }
        }
        myPos[(cur ^ 1) * 16 + l15] = rel_of(raw_n);  // (the other half of the double buffer: its readers finished a trip ago)
    }

    // ---- workgroup record: [C][G+4] = gM row, ga.xyz, gb; then gcW
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        accAB[u].x += __shfl_xor(accAB[u].x, 16, WAVE); accAB[u].x += __shfl_xor(accAB[u].x, 32, WAVE);
        accAB[u].y += __shfl_xor(accAB[u].y, 16, WAVE); accAB[u].y += __shfl_xor(accAB[u].y, 32, WAVE);
        accAB[u].z += __shfl_xor(accAB[u].z, 16, WAVE); accAB[u].z += __shfl_xor(accAB[u].z, 32, WAVE);
        accAB[u].w += __shfl_xor(accAB[u].w, 16, WAVE); accAB[u].w += __shfl_xor(accAB[u].w, 32, WAVE);
    }
#pragma unroll
    for (int t = 0; t < GT; ++t) { tcw[t] += __shfl_xor(tcw[t], 16, WAVE); tcw[t] += __shfl_xor(tcw[t], 32, WAVE); }
    float *rec = part + (size_t)blockIdx.x * K::REC;
    if (PW == 1) {  // every wavefront owns its channels: straight to the record
#pragma unroll
        for (int u = 0; u < UT; ++u) {
#pragma unroll
            for (int t = 0; t < GT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = 16 * t + l15, ch = c0 + 16 * u + 4 * q + r;
                    if (g < G) rec[(size_t)ch * PER + g] = accM[u][t][r];
                }
            if (q == 0) {
                float *d = rec + (size_t)(c0 + 16 * u + l15) * PER + G;
                d[0] = accAB[u].x; d[1] = accAB[u].y; d[2] = accAB[u].z; d[3] = accAB[u].w;
            }
        }
        if (sub == 0 && q == 0) {
#pragma unroll
            for (int t = 0; t < GT; ++t) rec[(size_t)C * PER + 16 * t + l15] = tcw[t];
        }
    } else {  // PW point slots share the channels: add up in LDS, one wavefront after the other (fixed order)
        __syncthreads();
        for (int turn = 0; turn < 4; ++turn) {
            if (wid == turn) {
#pragma unroll
                for (int u = 0; u < UT; ++u) {
#pragma unroll
                    for (int t = 0; t < GT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int g = 16 * t + l15, ch = c0 + 16 * u + 4 * q + r;
                            if (g < G) sFin[ch * PER + g] += accM[u][t][r];
                        }
                    if (q == 0) {
                        float *d = sFin + (c0 + 16 * u + l15) * PER + G;
                        d[0] += accAB[u].x; d[1] += accAB[u].y; d[2] += accAB[u].z; d[3] += accAB[u].w;
                    }
                }
                if (sub == 0 && q == 0) {
#pragma unroll
                    for (int t = 0; t < GT; ++t) sFin[C * PER + 16 * t + l15] += tcw[t];
                }
            }
            __syncthreads();
        }
        for (int e = tid; e < (int)K::REC; e += 256) rec[e] = sFin[e];
    }
}

template <int G, int C, int NW>
int launch_logits_bwd_fused(int n, int k, const float *a, const float *b, const float *M, const float *coord, const int *idx,
                            const float *W1, const float *gW1, const double *gT1, const double *gT2, const FoldWBwdArgs &F,
                            float *gWt, float *part, size_t part_floats_avail, float *gM, float *ga, float *gb, float *gcW,
                            hipStream_t st) {
    using K = LogitsBwdCfg<G, C, NW>;
    const size_t lds = sizeof(float4) * (C + 4 * 2 * 16) + sizeof(float) * (2 * K::GT * 16 + (K::PW > 1 ? K::REC : 0));
    auto kern = logits_bwd_fused_kernel<G, C, NW>;
    // exactly the co-resident workgroups (every one stages its constants once, then walks its points; as gva_bwd_point)
    static int resident = 0;
    if (!resident) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int occ = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, 256, lds) != hipSuccess || occ < 1) occ = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident = std::max(64, std::min(occ * cus, 512));
    }
    const long long groups = ((long long)n + K::PW - 1) / K::PW;
    long long cap = std::min<long long>(resident, (long long)(part_floats_avail / K::REC));
    if (cap < 1) return PTV2_ERR_WORKSPACE;
    const int nblk = (int)std::max<long long>(1, std::min<long long>(groups, cap));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, n, k, a, b, M, coord, idx, W1, gW1, gT1, gT2, gWt, part, F);
    launch_finalize(st, (const float *)part, nblk, (int)K::REC, MapLogitsFused{gM, ga, gb, gcW, G, C});
    return PTV2_OK;
}

// ---- G = 6 (the full-resolution level, C = 48; k = 16): the same launch for 24-byte rows --------------------------------
// A point's W1 / gW1 block is 16 x 6 floats = 24 float4: lanes 0..23 load it in one coalesced request each, form gWt in the
// loaded layout (the same fused multiply-add: bit-identical to the rows kernel), store it for the gather kernel and drop
// it into a wave-private LDS record from which every lane picks its two operand layouts.  The contraction index of the
// D product is mapped g = (lane >> 4) + 4 j, so the six groups take two MFMAs (j = 0, 1) instead of the four a 16-wide g
// tile costs.  One wavefront per point; the four wavefronts of a workgroup add their registers up in LDS at the end.
template <int C>
__global__ __launch_bounds__(256) void logits_bwd_fused6_kernel(int n, const float *__restrict__ a, const float *__restrict__ b,
                                                                const float *__restrict__ M, const float *__restrict__ coord,
                                                                const int *__restrict__ idx, const float *__restrict__ W1,
                                                                const float *__restrict__ gW1, const double *__restrict__ gT1,
                                                                const double *__restrict__ gT2, float *__restrict__ gWt,
                                                                float *__restrict__ part, FoldWBwdArgs F) {
    constexpr int G = 6, UT = C / 16, PER = G + 4, REC = C * PER + 16, ROW = 16 * G;
    static_assert(C % 16 == 0, "a wavefront owns whole 16-channel tiles");
    extern __shared__ float4 lds4[];
    float4 *sAB = lds4;                          // [C] (a.xyz, b)
    float4 *sPos = sAB + C;                      // [4 waves][2][16] wave-private position records
    float *sC1 = (float *)(sPos + 4 * 2 * 16);   // [16]
    float *sC2 = sC1 + 16;                       // [16]
    float *sRow = sC2 + 16;                      // [4 waves][ROW] gWt of the wavefront's current point
    float *sFin = sRow + 4 * ROW;                // [REC]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;

    for (int ch = tid; ch < C; ch += 256) sAB[ch] = make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]);
    if (tid < 16) {  // c1 = gT1, c2 = 2 gT2: the two statistics paths of BN_w folded into every row gradient
        float c1 = 0.f, c2 = 0.f;
        if (tid < G) {
            if (F.gsc) {
                double t1, t2;
                float gg, gb_;
                fold_w_bwd_channel(F, tid, t1, t2, gg, gb_);
                c1 = (float)t1;
                c2 = 2.f * (float)t2;
                if (blockIdx.x == 0) { F.ggamma[tid] = gg; F.gbeta[tid] = gb_; }
            } else {
                c1 = (float)gT1[tid];
                c2 = 2.f * (float)gT2[tid];
            }
        }
        sC1[tid] = c1;
        sC2[tid] = c2;
    }
    for (int e = tid; e < REC; e += 256) sFin[e] = 0.f;
    __syncthreads();

    // folding constants of the loader layout: element e of float4 number `lane` is group (4 lane + e) % 6
    float c1L[4], c2L[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { c1L[e] = sC1[(4 * lane + e) % G]; c2L[e] = sC2[(4 * lane + e) % G]; }
    // M fragments (B operand of the D product): channel 16 u + l15, group q + 4 j
    float mreg[UT][2];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        mreg[u][0] = M[(size_t)(16 * u + l15) * G + q];
        mreg[u][1] = q < 2 ? M[(size_t)(16 * u + l15) * G + 4 + q] : 0.f;
    }
    float4 abr[UT];  // (a, b) of my channels: loop-invariant, kept out of the loop's LDS round trips
#pragma unroll
    for (int u = 0; u < UT; ++u) abr[u] = sAB[16 * u + l15];
    float4 accAB[UT];
    v4f accM[UT];
    float tcw = 0.f;
#pragma unroll
    for (int u = 0; u < UT; ++u) { accAB[u] = make_float4(0.f, 0.f, 0.f, 0.f); accM[u] = (v4f){0.f, 0.f, 0.f, 0.f}; }

    const long long stride = (long long)gridDim.x * 4;
    float4 *myPos = sPos + wid * 32;
    float *myRow = sRow + wid * ROW;
    // Every load of the loop is UNCONDITIONAL (indices clamped to something valid, the value masked afterwards): a load
    // inside a divergent `if` splits the loop into basic blocks, and the compiler then waits with vmcnt(0) at every join --
    // the next point's rows, requested a few instructions earlier, were waited for on the spot (2.7 us per point).
    const long long last = (long long)n - 1;
    const int l16 = lane & 15, l24 = lane < ROW / 4 ? lane : ROW / 4 - 1;
    // (and a value that is only used under a divergent condition gets its load sunk into that branch: the four lane
    // quarters therefore all carry slot l16's id and position and all write the -- identical -- position record)
    auto load_idx = [&](long long pt) -> int {  // beyond the end: the last point's ids (valid, never used)
        return idx[(pt < n ? pt : last) * 16 + l16];
    };
    struct Raw { float sx, sy, sz, px, py, pz; bool ok; };
    auto load_raw = [&](long long pt, int src) -> Raw {
        const long long ps = src >= 0 ? src : 0, pp = pt < n ? pt : last;
        Raw r;
        r.sx = coord[3 * ps]; r.sy = coord[3 * ps + 1]; r.sz = coord[3 * ps + 2];
        r.px = coord[3 * pp]; r.py = coord[3 * pp + 1]; r.pz = coord[3 * pp + 2];
        r.ok = pt < n && src >= 0;
        return r;
    };
    auto rel_of = [](const Raw &r) { return r.ok ? make_float4(r.sx - r.px, r.sy - r.py, r.sz - r.pz, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f); };
    float4 wn, gn;
    auto load_rows = [&](long long pt) {
        const size_t o = (size_t)(pt < n ? pt : last) * ROW + 4 * l24;
        wn = *(const float4 *)(W1 + o);
        gn = *(const float4 *)(gW1 + o);
    };
    const long long pt0 = (long long)blockIdx.x * 4 + wid;
    load_rows(pt0);
    int idx_n = load_idx(pt0 + stride);
    {
        const float4 r0 = rel_of(load_raw(pt0, load_idx(pt0)));
        myPos[l16] = r0;
    }
    int cur = 0;
    for (long long pt = pt0; pt < n; pt += stride, cur ^= 1) {  // (no workgroup barrier inside: trip counts differ per wavefront)
        // ---- gWt of this point in the loaded layout: to the gather kernel's tensor and to the wave-private record
        const float4 u4 = make_float4(__builtin_fmaf(wn.x, c2L[0], gn.x + c1L[0]), __builtin_fmaf(wn.y, c2L[1], gn.y + c1L[1]),
                                      __builtin_fmaf(wn.z, c2L[2], gn.z + c1L[2]), __builtin_fmaf(wn.w, c2L[3], gn.w + c1L[3]));
        if (lane < ROW / 4) {
            *(float4 *)(gWt + (size_t)pt * ROW + 4 * lane) = u4;
            *(float4 *)(myRow + 4 * lane) = u4;
        }
        // ---- requests for the next points
        load_rows(pt + stride);
        const Raw raw_n = load_raw(pt + stride, idx_n);
        idx_n = load_idx(pt + 2 * stride);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the gWt record and position record `cur` of this wavefront are in LDS
        // ---- the two operand layouts
        const float uA0 = myRow[l15 * G + q];                       // (slot l15, group q + 4 j)
        const float uA1r = myRow[l15 * G + 4 + (q & 1)];
        const float uA1 = q < 2 ? uA1r : 0.f;
        float uB[4];                                                // (slot 4 st + q, group l15)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float t = myRow[(4 * st + q) * G + (l15 < G ? l15 : 0)];
            uB[st] = l15 < G ? t : 0.f;
            tcw += uB[st];
        }
        float4 rp[4], pq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { rp[r] = myPos[cur * 16 + 4 * q + r]; pq[r] = myPos[cur * 16 + 4 * r + q]; }
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const float4 ab = abr[u];
            v4f d = (v4f){0.f, 0.f, 0.f, 0.f};
            d = mfma4l(uA0, mreg[u][0], d);  // D (s, ch) = gWt (s, g) M^T (g, ch)
            d = mfma4l(uA1, mreg[u][1], d);
#pragma unroll
            for (int st = 0; st < 4; ++st) {  // gM (ch, g) += P^T gWt: contraction over the slots s = 4 st + q
                const float P = pe_act(ab.x, ab.y, ab.z, ab.w, pq[st].x, pq[st].y, pq[st].z);
                accM[u] = mfma4l(P, uB[st], accM[u]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // rows s = 4 q + r of column ch
                const float P = pe_act(ab.x, ab.y, ab.z, ab.w, rp[r].x, rp[r].y, rp[r].z);
                const float gpre = P > 0.f ? d[r] : 0.f;
                accAB[u].x = __builtin_fmaf(gpre, rp[r].x, accAB[u].x);
                accAB[u].y = __builtin_fmaf(gpre, rp[r].y, accAB[u].y);
                accAB[u].z = __builtin_fmaf(gpre, rp[r].z, accAB[u].z);
                accAB[u].w += gpre;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();  // every read of the two records is done before they are written again
        myPos[(cur ^ 1) * 16 + l16] = rel_of(raw_n);
    }

    // ---- workgroup record: [C][G+4] = gM row, ga.xyz, gb; then gcW -- the four wavefronts in turn (fixed order)
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        accAB[u].x += __shfl_xor(accAB[u].x, 16, WAVE); accAB[u].x += __shfl_xor(accAB[u].x, 32, WAVE);
        accAB[u].y += __shfl_xor(accAB[u].y, 16, WAVE); accAB[u].y += __shfl_xor(accAB[u].y, 32, WAVE);
        accAB[u].z += __shfl_xor(accAB[u].z, 16, WAVE); accAB[u].z += __shfl_xor(accAB[u].z, 32, WAVE);
        accAB[u].w += __shfl_xor(accAB[u].w, 16, WAVE); accAB[u].w += __shfl_xor(accAB[u].w, 32, WAVE);
    }
    tcw += __shfl_xor(tcw, 16, WAVE);
    tcw += __shfl_xor(tcw, 32, WAVE);
    __syncthreads();
    for (int turn = 0; turn < 4; ++turn) {
        if (wid == turn) {
#pragma unroll
            for (int u = 0; u < UT; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (l15 < G) sFin[(16 * u + 4 * q + r) * PER + l15] += accM[u][r];
                if (q == 0) {
                    float *d = sFin + (16 * u + l15) * PER + G;
                    d[0] += accAB[u].x; d[1] += accAB[u].y; d[2] += accAB[u].z; d[3] += accAB[u].w;
                }
            }
            if (q == 0) sFin[C * PER + l15] += tcw;
        }
        __syncthreads();
    }
    float *rec = part + (size_t)blockIdx.x * REC;
    for (int e = tid; e < REC; e += 256) rec[e] = sFin[e];
}

template <int C>
int launch_logits_bwd_fused6(int n, const float *a, const float *b, const float *M, const float *coord, const int *idx,
                             const float *W1, const float *gW1, const double *gT1, const double *gT2, const FoldWBwdArgs &F,
                             float *gWt, float *part, size_t part_floats_avail, float *gM, float *ga, float *gb, float *gcW,
                             hipStream_t st) {
    constexpr int REC = C * 10 + 16;
    const size_t lds = sizeof(float4) * (C + 4 * 2 * 16) + sizeof(float) * (2 * 16 + 4 * 96 + REC);
    auto kern = logits_bwd_fused6_kernel<C>;
    static int resident = 0;  // exactly the co-resident workgroups, as launch_logits_bwd_fused
    if (!resident) {
        int occ = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, 256, lds) != hipSuccess || occ < 1) occ = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident = std::max(64, std::min(occ * cus, 1024));
    }
    const long long groups = ((long long)n + 3) / 4;
    long long cap = std::min<long long>(resident, (long long)(part_floats_avail / REC));
    if (cap < 1) return PTV2_ERR_WORKSPACE;
    const int nblk = (int)std::max<long long>(1, std::min<long long>(groups, cap));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, n, a, b, M, coord, idx, W1, gW1, gT1, gT2, gWt, part, F);
    launch_finalize(st, (const float *)part, nblk, REC, MapLogitsFused{gM, ga, gb, gcW, 6, C});
    return PTV2_OK;
}

}  // namespace gva

// 1 when (k, c, g) has an instantiation of the fused rows + parameter-gradient kernel
int gva_logits_bwd_fused_supported(int k, int c, int g) {
    if (k < 1 || k > 16) return 0;
    // (64, 512) -- the ScanNet cfg's deepest level, a few dozen points -- would need 128 accumulator registers per lane and
    // spills: it stays on the staged kernels of gva_bwd.hip
    if (g == 6 && c == 48) {  // (24-byte rows: float4 loads of a point's block need k = 16)
        return k == 16;
    }
    return (g == 12 && c == 96) || (g == 24 && c == 192) || (g == 48 && c == 384);
}

int gva_logits_bwd_fused_launch(int n, int k, int c, int g, const float *a, const float *b, const float *M, const float *coord,
                                const int *idx, const float *W1, const float *gW1, const double *gT1, const double *gT2,
                                const gva::FoldWBwdArgs &F, float *gWt, float *part, size_t part_floats_avail, float *gM, float *ga,
                                float *gb, float *gcW, hipStream_t st) {
    using namespace gva;
#define ARGS n, k, a, b, M, coord, idx, W1, gW1, gT1, gT2, F, gWt, part, part_floats_avail, gM, ga, gb, gcW, st
    if (g == 6 && c == 48)
        return launch_logits_bwd_fused6<48>(n, a, b, M, coord, idx, W1, gW1, gT1, gT2, F, gWt, part, part_floats_avail, gM, ga, gb, gcW, st);
    if (g == 12 && c == 96) return launch_logits_bwd_fused<12, 96, 1>(ARGS);
    if (g == 24 && c == 192) return launch_logits_bwd_fused<24, 192, 4>(ARGS);  // (2 waves per point: 306 registers, 1 wave / SIMD)
    if (g == 48 && c == 384) return launch_logits_bwd_fused<48, 384, 4>(ARGS);
#undef ARGS
    return PTV2_ERR_ARG;
}

// ============================================================================ forward ==
// W1 (s,g) = P (s,ch) M (ch,g) + kW[idx_s] - qW + cW per point on the matrix cores, plus the column sums T1, T2 that BN_w
// needs -- the pipelined form of attention_logits_point_kernel (gva_bwd_point.hip), used from G = 12 up.
// Round 2's forms at the deep levels: a flat one-lane-per-slot kernel (G = 12, 24: every workgroup of 64 rows staged the
// whole C x G matrix M in LDS first, 24 MB of L2 reads at 4.5 k points, then ran 1 344 vector instructions per wavefront:
// 41 us) and the point kernel (G = 48: its M fragments were requested inside the reduction loop, four steps ahead:
// 24 dependent L2 round trips per point, 44 us for 1 074 points).
// Here the M fragments of a wavefront's channels live in registers for the whole launch, a wavefront walks its points with
// the next point's neighbour coordinates (ids two points ahead) and the current point's kW rows / qW row in flight behind
// the products, and nothing in the loop synchronises more than the NW wavefronts that share a point.
namespace gva {

template <int G, int C, int NW>
__global__ __launch_bounds__(256) void logits_fwd_mfma_kernel(int n, int k, const float *__restrict__ kW,
                                                              const float *__restrict__ qW, const float *__restrict__ a,
                                                              const float *__restrict__ b, const float *__restrict__ M,
                                                              const float *__restrict__ cW, const float *__restrict__ coord,
                                                              const int *__restrict__ idx, float *__restrict__ W1, float *part,
                                                              unsigned *counter, double *__restrict__ T1,
                                                              double *__restrict__ T2, FoldWFwdArgs F) {
    constexpr int GT = (G + 15) / 16, G16 = GT * 16, PW = 4 / NW, CW = C / NW, KS = CW / 4;
    constexpr int FT = (GT + NW - 1) / NW;  // g tiles a wavefront finishes (tile tg belongs to wavefront tg % NW of the point)
    extern __shared__ float4 lds4[];
    float4 *sAB = lds4;                                  // [C]
    float4 *sPos = sAB + C;                              // [4 waves][2][16]
    int *sSrc = (int *)(sPos + 4 * 2 * 16);              // [4 waves][2][16]
    float *sRed = (float *)(sSrc + 4 * 2 * 16);          // [PW][NW][GT][4][64]   (NW > 1 only)
    __shared__ float s_w[4][2 * G16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = wid / NW, sub = wid % NW;
    const int l15 = lane & 15, q = lane >> 4;
    const int c0 = sub * CW;
    for (int ch = tid; ch < C; ch += 256) sAB[ch] = make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]);
    // B operand: M[ch = c0 + 4 ks + q][g = 16 tg + l15], this wavefront's channels, resident
    float mreg[KS][GT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int tg = 0; tg < GT; ++tg) {
            const int g = 16 * tg + l15;
            mreg[ks][tg] = g < G ? M[(size_t)(c0 + 4 * ks + q) * G + g] : 0.f;
        }
    float t1[FT], t2[FT], cw[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        const int g = 16 * (f * NW + sub) + l15;
        t1[f] = t2[f] = 0.f;
        cw[f] = (f * NW + sub < GT && g < G) ? cW[g] : 0.f;
    }
    __syncthreads();

    const long long stride = (long long)gridDim.x * PW;
    float4 *myPos = sPos + wid * 32;
    int *mySrc = sSrc + wid * 32;
    // (every load of the loop is unconditional, every lane quarter carries slot l15's id and position: see ptv2_zero_pad in
    // common.h and logits_bwd_fused6_kernel above)
    const long long lastp = (long long)n - 1;
    const int lk = l15 < k ? l15 : 0;
    auto load_idx = [&](long long pt) -> int { return idx[(pt < n ? pt : lastp) * k + lk]; };
    struct Raw { float sx, sy, sz, px, py, pz; bool okl, oks; int src; };
    auto load_raw = [&](long long pt, int src) -> Raw {
        const long long ps = src >= 0 ? src : 0, pp = pt < n ? pt : lastp;
        Raw r;
        r.sx = coord[3 * ps]; r.sy = coord[3 * ps + 1]; r.sz = coord[3 * ps + 2];
        r.px = coord[3 * pp]; r.py = coord[3 * pp + 1]; r.pz = coord[3 * pp + 2];
        r.okl = pt < n && l15 < k;
        r.oks = r.okl && src >= 0;
        r.src = src;
        return r;
    };
    auto rel_of = [](const Raw &r) { return r.oks ? make_float4(r.sx - r.px, r.sy - r.py, r.sz - r.pz, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f); };
    const long long pt0 = (long long)blockIdx.x * PW + p;
    int idx_n = load_idx(pt0 + stride);
    {
        const Raw r0 = load_raw(pt0, load_idx(pt0));
        myPos[l15] = rel_of(r0);
        mySrc[l15] = r0.okl ? r0.src : -1;
    }
    int cur = 0;
    // (NW > 1: the wavefronts of a point meet at a workgroup barrier, so every wavefront runs the workgroup's trip count)
    const long long base0 = (long long)blockIdx.x * PW;
    for (long long base = base0; base < n; base += stride, cur ^= 1) {
        const long long pt = base + p;
        const bool act = pt < n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float4 myp = myPos[cur * 16 + l15];  // slot l15 of my point: the A operand's row
        // requests: this point's kW rows and qW row (consumed after the products), the next point's coordinates, ids two ahead
        float kv[FT][4], qv[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            const int tg = f * NW + sub, g = 16 * tg + l15;
            const bool gok = tg < GT && g < G && act;
            qv[f] = ptv2_ld_or_zero(qW + pt * G + g, gok);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = 4 * q + r;
                const int src = mySrc[cur * 16 + s];
                kv[f][r] = ptv2_ld_or_zero(kW + (long long)src * G + g, gok && s < k && src >= 0);
            }
        }
        const Raw raw_n = load_raw(pt + stride, idx_n);
        idx_n = load_idx(pt + 2 * stride);
        // products: my channels
        v4f acc[GT];
#pragma unroll
        for (int tg = 0; tg < GT; ++tg) acc[tg] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 ab = sAB[c0 + 4 * ks + q];
            const float P = pe_act(ab.x, ab.y, ab.z, ab.w, myp.x, myp.y, myp.z);
#pragma unroll
            for (int tg = 0; tg < GT; ++tg) acc[tg] = mfma4l(P, mreg[ks][tg], acc[tg]);
        }
        if (NW > 1) {  // channel parts -> LDS, the wavefront that owns a g tile adds them up (fixed order)
#pragma unroll
            for (int tg = 0; tg < GT; ++tg)
#pragma unroll
                for (int r = 0; r < 4; ++r) sRed[((((size_t)p * NW + sub) * GT + tg) * 4 + r) * 64 + lane] = acc[tg][r];
            __syncthreads();
        }
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            const int tg = f * NW + sub, g = 16 * tg + l15;
            if (tg < GT) {  // wave-uniform
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (NW > 1) {
                        float t = 0.f;
#pragma unroll
                        for (int w = 0; w < NW; ++w) t += sRed[((((size_t)p * NW + w) * GT + tg) * 4 + r) * 64 + lane];
                        v[r] = t;
                    } else {
                        float t = 0.f;
#pragma unroll
                        for (int tt = 0; tt < GT; ++tt) t = tt == tg ? acc[tt][r] : t;
                        v[r] = t;
                    }
                }
                // (the sums are formed OUTSIDE the store's condition: with every use of the kW / qW rows under `act && g < G` the
                // compiler sank their loads into that branch, behind the products -- the gather was requested and waited for at
                // the end of the trip, ~1 us per point exposed)
                const bool okg = act && g < G;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s = 4 * q + r;
                    const bool ok = okg && s < k;
                    const float val = v[r] + (kv[f][r] - qv[f]) + cw[f];
                    t1[f] += ok ? val : 0.f;
                    t2[f] = ok ? __builtin_fmaf(val, val, t2[f]) : t2[f];
                    if (ok) W1[(pt * k + s) * G + g] = val;
                }
            }
        }
        myPos[(cur ^ 1) * 16 + l15] = rel_of(raw_n);  // (all four lane quarters: identical values)
        mySrc[(cur ^ 1) * 16 + l15] = raw_n.okl ? raw_n.src : -1;
        if (NW > 1) __syncthreads();  // sRed is rewritten by the next trip
    }
    // workgroup record [T1 (G) | T2 (G)]
    for (int e = tid; e < 4 * 2 * G16; e += 256) (&s_w[0][0])[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        t1[f] += __shfl_xor(t1[f], 16, WAVE); t1[f] += __shfl_xor(t1[f], 32, WAVE);
        t2[f] += __shfl_xor(t2[f], 16, WAVE); t2[f] += __shfl_xor(t2[f], 32, WAVE);
        const int tg = f * NW + sub;
        if (q == 0 && tg < GT) { s_w[wid][16 * tg + l15] = t1[f]; s_w[wid][G16 + 16 * tg + l15] = t2[f]; }
    }
    __syncthreads();
    if (tid < 2 * G) {
        const int col = tid < G ? tid : G16 + (tid - G);
        float v = 0.f;
        for (int wv = 0; wv < 4; ++wv) v += s_w[wv][col];
        part_store(part + (size_t)blockIdx.x * 2 * G + tid, v);
    }
    if (counter && last_block_arrives(counter)) finalize_logit_sums(part, gridDim.x, G, T1, T2, F);
}

template <int G, int C, int NW>
int launch_logits_fwd_mfma(int n, int k, const float *kW, const float *qW, const float *a, const float *b, const float *M,
                           const float *cW, const float *coord, const int *idx, float *W1, float *part, double *T1, double *T2,
                           const FoldWFwdArgs &F, hipStream_t st) {
    constexpr int GT = (G + 15) / 16, PW = 4 / NW;
    const size_t lds = sizeof(float4) * (C + 4 * 2 * 16) + sizeof(int) * 4 * 2 * 16 +
                       (NW > 1 ? sizeof(float) * (size_t)PW * NW * GT * 4 * 64 : 0);
    auto kern = logits_fwd_mfma_kernel<G, C, NW>;
    static int resident = 0;
    if (!resident) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int occ = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, 256, lds) != hipSuccess || occ < 1) occ = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident = std::max(64, std::min(occ * cus, MAX_BLOCKS));
    }
    const long long groups = ((long long)n + PW - 1) / PW;
    const int nblk = (int)std::max<long long>(1, std::min<long long>(groups, resident));
    const bool own_final = (size_t)nblk * 2 * G <= FUSED_FINAL_MAX;
    unsigned *cnt = own_final ? ptv2_stream_counters(st) : nullptr;
    if (own_final && !cnt) return PTV2_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, n, k, kW, qW, a, b, M, cW, coord, idx, W1, part,
                       cnt ? cnt + CNT_LOGITS_FWD : nullptr, T1, T2, F);
    if (!own_final)
        hipLaunchKernelGGL(finalize_logit_sums_kernel, dim3((G + FLS_GROUPS - 1) / FLS_GROUPS), dim3(1024), 0, st, (const float *)part,
                           nblk, G, T1, T2, F);
    return PTV2_OK;
}

}  // namespace gva

int gva_logits_fwd_mfma_supported(int k, int c, int g) {
    if (k < 1 || k > 16) return 0;
    // (6, 48) stays on the flat one-lane-per-slot kernel: the MFMA form measured equal there (58 vs 59 us at 120 k points: with
    // 6 of 16 tile columns in use the epilogue stores 24-byte segments from 6 lanes) and is not instantiated
    return (g == 12 && c == 96) || (g == 24 && c == 192) || (g == 48 && c == 384);
}

// part: >= MAX_BLOCKS * 2 g floats
int gva_logits_fwd_mfma_launch(int n, int k, int c, int g, const float *kW, const float *qW, const float *a, const float *b,
                               const float *M, const float *cW, const float *coord, const int *idx, float *W1, float *part,
                               double *T1, double *T2, const gva::FoldWFwdArgs &F, hipStream_t st) {
    using namespace gva;
#define ARGS n, k, kW, qW, a, b, M, cW, coord, idx, W1, part, T1, T2, F, st
    if (g == 12 && c == 96) return launch_logits_fwd_mfma<12, 96, 1>(ARGS);
    if (g == 24 && c == 192) return launch_logits_fwd_mfma<24, 192, 1>(ARGS);
    if (g == 48 && c == 384) return launch_logits_fwd_mfma<48, 384, 4>(ARGS);
#undef ARGS
    return PTV2_ERR_ARG;
}
