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
//   aggregate backward one wavefront per point: recompute y, softmax; grad w from the v path, the
//                     positional path (g_A) and g_sw; softmax / Linear(G,G) / ReLU / BN_w-affine
//                     backward; writes gW1 and w; accumulates grad sc, sh, Ww2, bw2, a, b
//                     gather kernel grad v (via inverse table)
#include <algorithm>

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
template <int G>
__global__ __launch_bounds__(TPB) void logits_bwd_rows_kernel(long long rows, const float *__restrict__ W1,
                                                              const float *__restrict__ gW1,
                                                              const double *__restrict__ gT1,
                                                              const double *__restrict__ gT2, float *__restrict__ gWt,
                                                              float *__restrict__ part) {
    __shared__ float s_w[WPB][G];
    float t[G], c1[G], c2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { t[g] = 0.f; c1[g] = (float)gT1[g]; c2[g] = 2.f * (float)gT2[g]; }
    for (long long row = (long long)blockIdx.x * TPB + threadIdx.x; row < rows; row += (long long)gridDim.x * TPB) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float v = __builtin_fmaf(W1[row * G + g], c2[g], gW1[row * G + g] + c1[g]);
            gWt[row * G + g] = v;
            t[g] += v;
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
        part[(size_t)blockIdx.x * G + threadIdx.x] = v;
    }
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
        for (int s = 0; s < k; ++s) q += gWt[((long long)j * k + s) * g + gi];
        gqW[e] = -q;
        if (inv_ptr) {
            float acc = 0.f;
            for (int p = inv_ptr[j]; p < inv_ptr[j + 1]; ++p) acc += gWt[(long long)inv_rows[p] * g + gi];
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

// final: partials [nblk][c][G+4] -> gM (c,G), ga (c,3), gb (c)
struct MapLogitsParams {
    float *gM, *ga, *gb;
    int g;
    __device__ void operator()(int e, double v) const {
        const int per = g + 4, ch = e / per, j = e - ch * per;
        if (j < g) gM[ch * g + j] = (float)v;
        else if (j < g + 3) ga[ch * 3 + (j - g)] = (float)v;
        else gb[ch] = (float)v;
    }
};

// ========================================================= aggregate backward ==
// one wavefront (= one 64-thread workgroup) per point; all LDS images are private to the wave.
struct BwdLds {
    __host__ __device__ static constexpr size_t floats(int G, int K, int C) {
        const size_t GP = AggLds::gp(G), G4 = AggLds::G4(G);
        return AggLds::r4(2 * (size_t)G * GP + 3 * (size_t)G)      // Ww2, Ww2^T, bw2, sc, sh
               + 4 * (size_t)C                                       // (a, b)
               + AggLds::r4(5 * (size_t)K + 3 * (size_t)K * GP)      // pos, src, Y, Wt, GW
               + (size_t)K * G4                                      // W rows
               + AggLds::r4(64 * GP)                                 // g_A chunk tile
               + AggLds::r4(3 * (size_t)G + (size_t)G * G) + 4 * (size_t)C;  // accumulators
    }
};

template <int G>
__global__ __launch_bounds__(WAVE) void aggregate_bwd_kernel(
    int n, int k, int c, const float *__restrict__ W1, const float *__restrict__ sc, const float *__restrict__ sh,
    const float *__restrict__ Ww2, const float *__restrict__ bw2, const float *__restrict__ v,
    const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ coord,
    const int *__restrict__ idx, const float *__restrict__ g_out, const float *__restrict__ g_A,
    const float *__restrict__ g_sw, float *__restrict__ gW1, float *__restrict__ wbuf, float *gv_atomic,
    float *__restrict__ part) {
    extern __shared__ float4 lds4[];
    float *lds = (float *)lds4;
    constexpr int GP = AggLds::gp(G);
    constexpr int G4 = AggLds::G4(G);
    const int lane = threadIdx.x;
    float *sWw2 = lds;                    // [G][GP]  Ww2[g][g']
    float *sWw2T = sWw2 + G * GP;         // [G][GP]  Ww2[g][g'] stored at [g'][g]
    float *sBw2 = sWw2T + G * GP;
    float *sSc = sBw2 + G;
    float *sSh = sSc + G;
    float4 *sAB = (float4 *)(lds + AggLds::r4(2 * (size_t)G * GP + 3 * (size_t)G));
    float *pbase = (float *)(sAB + c);
    float4 *sPos = (float4 *)pbase;       // [K]
    int *sSrc = (int *)(pbase + 4 * k);   // [K]
    float *sY = pbase + 5 * k;            // [K][GP]
    float *sWt = sY + (size_t)k * GP;     // [K][GP] unmasked softmax; later: gW1 staging
    float *sGW = sWt + (size_t)k * GP;    // [K][GP] grad w, then grad z
    float *sW = pbase + AggLds::r4(5 * (size_t)k + 3 * (size_t)k * GP);  // [K][G4] masked softmax
    float *sT = sW + (size_t)k * G4;      // [64][GP] g_A chunk tile
    float *accS = sT + AggLds::r4(64 * GP);  // [3G]: gsc, gsh, gbw2 ; then [G*G] gWw2
    float *accW = accS + 3 * G;
    float *accAB = accS + AggLds::r4(3 * (size_t)G + (size_t)G * G);  // [C][4]

    for (int i = lane; i < G * G; i += WAVE) {
        const int g = i / G, gq = i - g * G;
        sWw2[g * GP + gq] = Ww2[i];
        sWw2T[gq * GP + g] = Ww2[i];
        accW[i] = 0.f;
    }
    for (int i = lane; i < G; i += WAVE) {
        sBw2[i] = bw2[i]; sSc[i] = sc[i]; sSh[i] = sh[i];
        accS[i] = accS[G + i] = accS[2 * G + i] = 0.f;
    }
    for (int i = lane; i < c; i += WAVE) {
        sAB[i] = make_float4(a[3 * i], a[3 * i + 1], a[3 * i + 2], b[i]);
        ((float4 *)accAB)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    const int I = c / G;
    const int items = G * k;
    const int J = WAVE / k;              // channel slices of the (s, j) mapping
    const int ms = lane & (k - 1), mj = lane / k;

    for (int pt = blockIdx.x; pt < n; pt += gridDim.x) {
        // 0: neighbour slots
        if (lane < k) {
            Rel r = rel_pos(coord, idx, (long long)pt * k + lane, pt);
            sPos[lane] = make_float4(r.x, r.y, r.z, r.src >= 0 ? 1.f : 0.f);
            sSrc[lane] = r.src;
        }
        // 1: y = ReLU(sc W1 + sh); grad-w accumulator starts at g_sw
        for (int item = lane; item < items; item += WAVE) {
            const int s = item / G, g = item - s * G;
            sY[s * GP + g] = fmaxf(__builtin_fmaf(sSc[g], W1[(long long)pt * items + item], sSh[g]), 0.f);
            sGW[s * GP + g] = g_sw[(long long)pt * G + g];
        }
        __syncthreads();
        // 2: softmax over s
        for (int base = 0; base < items; base += WAVE) {
            const int item = base + lane;
            const bool act = item < items;
            const int g = act ? item / k : 0, s = act ? item - g * k : 0;
            float z = sBw2[g];
            const float *yr = sY + s * GP, *wr = sWw2 + g * GP;
            for (int j = 0; j < G; ++j) z = __builtin_fmaf(yr[j], wr[j], z);
            float mx = z;
            for (int o = k >> 1; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, WAVE));
            const float e = expf(z - mx);
            float den = e;
            for (int o = k >> 1; o >= 1; o >>= 1) den += __shfl_xor(den, o, WAVE);
            const float wt = e / den;
            if (act) {
                sWt[s * GP + g] = wt;
                sW[s * G4 + g] = wt * sPos[s].w;
            }
        }
        __syncthreads();
        for (int item = lane; item < items; item += WAVE) {  // w (N,K,G) for the grad-v gather, coalesced
            const int s = item / G, g = item - s * G;
            wbuf[(long long)pt * items + item] = sW[s * G4 + g];
        }
        // 3: v path: grad w[s,g] += sum_{c in g} g_out[c] v[idx[s],c]
        for (int cb0 = 0; cb0 < c; cb0 += WAVE) {
            const int ch = cb0 + lane;
            const bool act = ch < c;
            const float go = act ? g_out[(long long)pt * c + ch] : 0.f;
            const int gl = act ? ch / I : 0;
            for (int s = 0; s < k; ++s) {
                const int src = sSrc[s];
                float val = 0.f;
                if (act && src >= 0) {
                    val = go * v[(long long)src * c + ch];
                    if (gv_atomic) atomicAdd(gv_atomic + (long long)src * c + ch, go * sW[s * G4 + gl]);
                }
                for (int o = I >> 1; o >= 1; o >>= 1) val += __shfl_xor(val, o, WAVE);
                if (act && (ch & (I - 1)) == 0) sGW[s * GP + gl] += val;
            }
        }
        __syncthreads();
        // 4: positional path through g_A, lanes = (slot s, channel slice j)
        {
            float wrow[G], acc[G];
            const float4 ps = sPos[ms];
#pragma unroll
            for (int g = 0; g < G; ++g) { wrow[g] = sW[ms * G4 + g]; acc[g] = 0.f; }
            for (int cb0 = 0; cb0 < c; cb0 += WAVE) {
                const int chl = cb0 + lane;
                if (chl < c) {
#pragma unroll
                    for (int g = 0; g < G; ++g) sT[lane * GP + g] = g_A[((long long)g * n + pt) * c + chl];
                }
                __syncthreads();
                const int cend = (c - cb0) < WAVE ? (c - cb0) : WAVE;
                for (int cl = mj; cl < WAVE; cl += J) {  // uniform trip count: shuffles below need every lane
                    const bool act = cl < cend;
                    float ga0 = 0.f, ga1 = 0.f, ga2 = 0.f, gb0 = 0.f;
                    if (act) {
                        const float4 ab = sAB[cb0 + cl];
                        const float P = pe_act(ab.x, ab.y, ab.z, ab.w, ps.x, ps.y, ps.z);
                        const float *tr = sT + cl * GP;
                        float gP = 0.f;
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const float t = tr[g];
                            gP = __builtin_fmaf(wrow[g], t, gP);
                            acc[g] = __builtin_fmaf(P, t, acc[g]);
                        }
                        const float gpre = P > 0.f ? gP : 0.f;
                        ga0 = gpre * ps.x; ga1 = gpre * ps.y; ga2 = gpre * ps.z; gb0 = gpre;
                    }
                    for (int o = k >> 1; o >= 1; o >>= 1) {  // sum over the k slots of this point
                        ga0 += __shfl_xor(ga0, o, WAVE); ga1 += __shfl_xor(ga1, o, WAVE);
                        ga2 += __shfl_xor(ga2, o, WAVE); gb0 += __shfl_xor(gb0, o, WAVE);
                    }
                    if (act && ms == 0) {
                        float4 *d = (float4 *)accAB + (cb0 + cl);
                        float4 cur = *d;
                        *d = make_float4(cur.x + ga0, cur.y + ga1, cur.z + ga2, cur.w + gb0);
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float t = acc[g];
                for (int o = WAVE >> 1; o >= k; o >>= 1) t += __shfl_xor(t, o, WAVE);
                if (mj == 0) sGW[ms * GP + g] += t;
            }
        }
        __syncthreads();
        // 5: softmax backward -> grad z (in sGW); grad bw2
        for (int base = 0; base < items; base += WAVE) {
            const int item = base + lane;
            const bool act = item < items;
            const int g = act ? item / k : 0, s = act ? item - g * k : 0;
            const float gw = act ? sGW[s * GP + g] * sPos[s].w : 0.f;
            const float wt = act ? sWt[s * GP + g] : 0.f;
            float dot = wt * gw;
            for (int o = k >> 1; o >= 1; o >>= 1) dot += __shfl_xor(dot, o, WAVE);
            const float gz = wt * (gw - dot);
            float tot = gz;
            for (int o = k >> 1; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, WAVE);
            if (act) {
                sGW[s * GP + g] = gz;
                if (s == 0) accS[2 * G + g] += tot;
            }
        }
        __syncthreads();
        // 6: Linear(G,G) / ReLU / BN_w-affine backward -> gW1 (staged in sWt), grad sc / sh
        for (int base = 0; base < items; base += WAVE) {
            const int item = base + lane;
            const bool act = item < items;
            const int gq = act ? item / k : 0, s = act ? item - gq * k : 0;
            float gy = 0.f;
            const float *zr = sGW + s * GP, *wt = sWw2T + gq * GP;
            for (int g = 0; g < G; ++g) gy = __builtin_fmaf(zr[g], wt[g], gy);
            const float y = sY[s * GP + gq];
            const float gu = (act && y > 0.f) ? gy : 0.f;
            const float u = act ? W1[((long long)pt * k + s) * G + gq] : 0.f;
            float t1 = gu, t2 = gu * u;
            for (int o = k >> 1; o >= 1; o >>= 1) { t1 += __shfl_xor(t1, o, WAVE); t2 += __shfl_xor(t2, o, WAVE); }
            if (act) {
                sWt[s * GP + gq] = sSc[gq] * gu;
                if (s == 0) { accS[gq] += t2; accS[G + gq] += t1; }
            }
        }
        // 7: grad Ww2[g][g'] += sum_s gz[s][g] y[s][g']
        for (int p = lane; p < G * G; p += WAVE) {
            const int g = p / G, gq = p - g * G;
            float t = 0.f;
            for (int s = 0; s < k; ++s) t = __builtin_fmaf(sGW[s * GP + g], sY[s * GP + gq], t);
            accW[p] += t;
        }
        __syncthreads();
        for (int item = lane; item < items; item += WAVE) {
            const int s = item / G, g = item - s * G;
            gW1[(long long)pt * items + item] = sWt[s * GP + g];
        }
        __syncthreads();
    }
    // per-wave partials: [gsc G][gsh G][gbw2 G][gWw2 G*G][ga,gb C*4]
    float *mypart = part + (size_t)blockIdx.x * (3 * G + G * G + 4 * (size_t)c);
    for (int i = lane; i < 3 * G; i += WAVE) mypart[i] = accS[i];
    for (int i = lane; i < G * G; i += WAVE) mypart[3 * G + i] = accW[i];
    for (int i = lane; i < 4 * c; i += WAVE) mypart[3 * G + G * G + i] = accAB[i];
}

struct MapAggParams {  // columns: [gsc G][gsh G][gbw2 G][gWw2 G*G][(ga,gb) C*4]
    float *gsc, *gsh, *gWw2, *gbw2, *ga, *gb;
    int g;
    __device__ void operator()(int e, double acc) const {
        const float v = (float)acc;
        if (e < g) gsc[e] = v;
        else if (e < 2 * g) gsh[e - g] = v;
        else if (e < 3 * g) gbw2[e - 2 * g] = v;
        else if (e < 3 * g + g * g) gWw2[e - 3 * g] = v;
        else {
            const int r = e - 3 * g - g * g, ch = r >> 2, j = r & 3;
            if (j < 3) ga[ch * 3 + j] = v; else gb[ch] = v;
        }
    }
};

// grad v[j,c] = sum over slots r that point at j of w[r, g(c)] * g_out[r / k, c]
__global__ __launch_bounds__(TPB) void aggregate_bwd_gv_kernel(int n, int k, int c, int g,
                                                               const float *__restrict__ wbuf,
                                                               const float *__restrict__ g_out,
                                                               const int *__restrict__ inv_ptr,
                                                               const int *__restrict__ inv_rows,
                                                               float *__restrict__ gv) {
    const int I = c / g;
    const long long total = (long long)n * c;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const int j = (int)(e / c), ch = (int)(e - (long long)j * c);
        const int gl = ch / I;
        float acc = 0.f;
        for (int p = inv_ptr[j]; p < inv_ptr[j + 1]; ++p) {
            const int r = inv_rows[p];
            acc = __builtin_fmaf(wbuf[(long long)r * g + gl], g_out[(long long)(r / k) * c + ch], acc);
        }
        gv[e] = acc;
    }
}

}  // namespace gva

using namespace gva;

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

extern "C" int gva_logits_backward_hip_launcher(int n, int k, int c, int g, const float *a, const float *b,
                                                const float *M, const float *coord, const int *idx,
                                                const float *W1, const float *gW1, const double *gT1,
                                                const double *gT2, const int *inv_ptr, const int *inv_rows,
                                                float *gkW, float *gqW, float *ga, float *gb, float *gM, float *gcW,
                                                void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 0 || k < 1 || c < 1 || g < 1) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < gva_workspace_bytes(n, k, c, g)) return PTV2_ERR_WORKSPACE;
    if (n == 0) return PTV2_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)n * k;
    float *part = (float *)workspace;
    float *gWt = (float *)((char *)workspace + rows_offset_bytes(c, g));
    const int nb_rows = stage_grid(rows, TPB * 2);
#define CALL(GG) \
    hipLaunchKernelGGL(logits_bwd_rows_kernel<GG>, dim3(nb_rows), dim3(TPB), 0, st, rows, W1, gW1, gT1, gT2, gWt, part)
    GVA_DISPATCH_G(g, CALL)
#undef CALL
    launch_finalize(st, (const float *)part, nb_rows, g, MapVec<float>{gcW});
    hipLaunchKernelGGL(logits_bwd_gather_kernel, dim3(stage_grid((long long)n * g, TPB)), dim3(TPB), 0, st, n, k, g,
                       (const float *)gWt, idx, inv_ptr, inv_rows, gkW, gqW);
    // params kernel: its partials go after the rows-kernel partials (still inside the partial region)
    float *ppart = part + (size_t)nb_rows * g;
    const int cbk = c < TPB ? c : TPB, nsl = c < TPB ? TPB / c : 1;
    const size_t comb_bytes = sizeof(float) * (size_t)nsl * cbk * (g + 4);
    const int nb_par = stage_grid(rows, PR_TILE * 2, MAX_PARAM_BLOCKS);
#define CALL(GG)                                                                                                   \
    if (comb_bytes > 32 * 1024)                                                                                    \
        (void)hipFuncSetAttribute((const void *)logits_bwd_params_kernel<GG>,                                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)comb_bytes);                    \
    hipLaunchKernelGGL(logits_bwd_params_kernel<GG>, dim3(nb_par), dim3(TPB), comb_bytes, st, n, k, c, a, b, M, coord, idx, \
                       (const float *)gWt, ppart)
    GVA_DISPATCH_G(g, CALL)
#undef CALL
    launch_finalize(st, (const float *)ppart, nb_par, c * (g + 4), MapLogitsParams{gM, ga, gb, g});
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int gva_aggregate_backward_hip_launcher(int n, int k, int c, int g, const float *W1, const float *sc,
                                                   const float *sh, const float *Ww2, const float *bw2,
                                                   const float *v, const float *a, const float *b,
                                                   const float *coord, const int *idx, const float *g_out,
                                                   const float *g_A, const float *g_sw, const int *inv_ptr,
                                                   const int *inv_rows, float *gW1, float *gsc, float *gsh,
                                                   float *gWw2, float *gbw2, float *gv, float *ga, float *gb,
                                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 0 || !pow2(k) || k > 64 || c < 1 || g < 1 || c % g != 0 || !pow2(c / g) || c / g > 64) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < gva_workspace_bytes(n, k, c, g)) return PTV2_ERR_WORKSPACE;
    if (n == 0) return PTV2_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_bytes = sizeof(float) * BwdLds::floats(g, k, c);
    if (lds_bytes > 160 * 1024) return PTV2_ERR_ARG;
    float *part = (float *)workspace;
    float *wbuf = (float *)((char *)workspace + rows_offset_bytes(c, g));
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds_bytes));
    const int nblk = std::min(n, std::min(256 * per_cu, (int)MAX_BLOCKS));
#define CALL(GG)                                                                                                    \
    if (lds_bytes > 32 * 1024)                                                                                      \
        (void)hipFuncSetAttribute((const void *)aggregate_bwd_kernel<GG>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds_bytes);                                                                  \
    hipLaunchKernelGGL(aggregate_bwd_kernel<GG>, dim3(nblk), dim3(WAVE), lds_bytes, st, n, k, c, W1, sc, sh, Ww2, bw2, v, a, \
                       b, coord, idx, g_out, g_A, g_sw, gW1, wbuf, inv_ptr ? (float *)nullptr : gv, part)
    GVA_DISPATCH_G(g, CALL)
#undef CALL
    const int len = 3 * g + g * g + 4 * c;
    launch_finalize(st, (const float *)part, nblk, len, MapAggParams{gsc, gsh, gWw2, gbw2, ga, gb, g});
    if (inv_ptr)
        hipLaunchKernelGGL(aggregate_bwd_gv_kernel, dim3(stage_grid((long long)n * c, TPB)), dim3(TPB), 0, st, n, k, c, g,
                           (const float *)wbuf, g_out, inv_ptr, inv_rows, gv);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
