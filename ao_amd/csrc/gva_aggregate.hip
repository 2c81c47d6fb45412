// ao_amd/csrc/gva_aggregate.hip -- softmax + aggregation stages of the fused grouped vector attention.
// Math: ao_amd/ptv2/gva.py.  Each stage is a flat, fully parallel kernel with ONE load phase, so that
// the chip hides gather latency with occupancy instead of serialising it per point:
//
//  forward
//   softmax_rows   one lane per neighbour slot (n,s): y = ReLU(sc*W1+sh) (G regs), z = y Ww2^T + bw2 with
//                  Ww2 as wave-uniform scalar operands, softmax over the K slots of a point through
//                  K-lane shuffles, mask.  W1 in, w (N,K,G) and sw (N,G) out -- one streaming pass.
//   aggregate_tile a workgroup takes a tile of points, stages their neighbour ids, relative positions and
//                  softmax rows in LDS once, then one lane per (point, channel):
//                  out_v = sum_s w v[idx] (row gathers, coalesced over channels),
//                  A[n,g,:] = sum_s w[s,g] ReLU(a.pos_s + b).
//  backward
//   bwd_tile       one wavefront per point: grad w = g_sw + <g_out, v[idx]> per group (8-lane shuffle
//                  reduce) + <P, g_A> (lanes = slot x channel-slice, g_A streamed through LDS in 64-channel
//                  chunks), and the folded-BN_p parameter gradients (ga, gb) as per-wave partials.
//   bwd_rows       one lane per slot: softmax backward, Linear(G,G) backward, ReLU, BN_w affine backward;
//                  writes gW1 and the two (N*K,G) operands (gz, y) of the Ww2 weight gradient, which is
//                  then the same split-K MFMA reduction as any Linear (dense.hip); gsc / gsh partials.
//   bwd_gv         grad v through the inverse neighbour table (fixed-order gather, no atomics).
#include <algorithm>
#include <cstdlib>

#include "gva_common.h"

namespace gva {

inline bool pow2(int k) { return k > 0 && (k & (k - 1)) == 0; }
__host__ __device__ constexpr int G4of(int G) { return (G + 3) & ~3; }
__host__ __device__ constexpr int GPof(int G) { return (G | 1) + ((G & 1) ? 2 : 0); }  // odd, >= G+1

template <int G>
__device__ __forceinline__ void load_row(const float *__restrict__ p, float (&v)[G]) {
    if (G % 4 == 0) {
#pragma unroll
        for (int g = 0; g < G; g += 4) {
            const float4 t = *(const float4 *)(p + g);
            v[g] = t.x; v[g + 1] = t.y; v[g + 2] = t.z; v[g + 3] = t.w;
        }
    } else if (G % 2 == 0) {
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            const float2 t = *(const float2 *)(p + g);
            v[g] = t.x; v[g + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) v[g] = p[g];
    }
}

template <int G>
__device__ __forceinline__ void store_row(float *__restrict__ p, const float (&v)[G]) {
    if (G % 4 == 0) {
#pragma unroll
        for (int g = 0; g < G; g += 4) *(float4 *)(p + g) = make_float4(v[g], v[g + 1], v[g + 2], v[g + 3]);
    } else if (G % 2 == 0) {
#pragma unroll
        for (int g = 0; g < G; g += 2) *(float2 *)(p + g) = make_float2(v[g], v[g + 1]);
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) p[g] = v[g];
    }
}

// z[g] = bw2[g] + sum_j y[j] Ww2[g][j]  (Ww2, bw2 wave-uniform), then softmax over the k lanes of a point
template <int G>
__device__ __forceinline__ void logits_softmax(const float (&y)[G], const float *Ww2, const float *bw2, int k,
                                               float (&wt)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float z = bw2[g];
#pragma unroll
        for (int j = 0; j < G; ++j) z = __builtin_fmaf(y[j], Ww2[g * G + j], z);
        wt[g] = z;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float mx = wt[g];
        for (int o = k >> 1; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, WAVE));
        const float e = expf(wt[g] - mx);
        float den = e;
        for (int o = k >> 1; o >= 1; o >>= 1) den += __shfl_xor(den, o, WAVE);
        wt[g] = e / den;
    }
}

// =================================================================== forward ==
template <int G, bool DROP>
__global__ __launch_bounds__(TPB) void softmax_rows_kernel(long long rows, int k, const float *__restrict__ W1,
                                                           const float *__restrict__ sc, const float *__restrict__ sh,
                                                           const float *__restrict__ Ww2,
                                                           const float *__restrict__ bw2, const int *__restrict__ idx,
                                                           float *__restrict__ w, float *__restrict__ sw, PtvDrop drop) {
    // wave-uniform operands from LDS (broadcast reads) instead of scalar loads: the G x G matrix does not fit the
    // scalar cache for G >= 24 and its miss latency dominated the deep-stage launches
    __shared__ float sWw2[G * G], sBw2[G], sSc[G], sSh[G];
    for (int i = threadIdx.x; i < G * G; i += TPB) sWw2[i] = Ww2[i];
    for (int i = threadIdx.x; i < G; i += TPB) { sBw2[i] = bw2[i]; sSc[i] = sc[i]; sSh[i] = sh[i]; }
    __syncthreads();
    const long long rows_pad = (rows + WAVE - 1) / WAVE * WAVE;  // whole waves: shuffles need every lane
    for (long long row = (long long)blockIdx.x * TPB + threadIdx.x; row < rows_pad; row += (long long)gridDim.x * TPB) {
        const bool act = row < rows;
        const long long r = act ? row : rows - 1;
        float y[G], wt[G];
        load_row<G>(W1 + r * G, y);
#pragma unroll
        for (int g = 0; g < G; ++g) y[g] = fmaxf(__builtin_fmaf(sSc[g], y[g], sSh[g]), 0.f);
        logits_softmax<G>(y, sWw2, sBw2, k, wt);
        const float valid = (act && idx[r] >= 0) ? 1.f : 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) wt[g] *= valid;
        if (DROP) {  // attention dropout on the softmax output (a template parameter: no registers for it when off)
#pragma unroll
            for (int g = 0; g < G; ++g) wt[g] *= ptv2_drop_factor(drop, (unsigned long long)r * G + g);
        }
        if (act) store_row<G>(w + r * G, wt);
#pragma unroll
        for (int g = 0; g < G; ++g)
            for (int o = k >> 1; o >= 1; o >>= 1) wt[g] += __shfl_xor(wt[g], o, WAVE);
        if (act && (r & (k - 1)) == 0) store_row<G>(sw + (r / k) * G, wt);
    }
}

// (measured and rejected, round 3: per-point LDS pitches of k + 1 position slots / k * G4 + 8 weight floats -- so that the two
// points a wavefront spans at C = 48 broadcast from different banks, the 13.6 % conflict share of round 2's counters -- and
// 8-byte weight reads at G = 6: 96 -> 103 us at 120 k points; the kernel is bound by its 16 row gathers per thread and the
// (N,G,C) store, not by LDS)
template <int G>
__global__ __launch_bounds__(TPB) void aggregate_tile_kernel(int n, int k, int c, int tp, const float *__restrict__ w,
                                                             const float *__restrict__ v, const float *__restrict__ a,
                                                             const float *__restrict__ b,
                                                             const float *__restrict__ coord,
                                                             const int *__restrict__ idx, float *__restrict__ out_v,
                                                             float *__restrict__ A) {
    extern __shared__ float4 lds4[];
    constexpr int G4 = G4of(G);
    float4 *sPos = lds4;                           // [tp*k]  (x,y,z,-)
    float *sW = (float *)(sPos + (size_t)tp * k);   // [tp*k][G4]
    int *sSrc = (int *)(sW + (size_t)tp * k * G4);  // [tp*k]
    const int n0 = blockIdx.x * tp;
    const int cnt = (n - n0) < tp ? (n - n0) : tp;
    for (int e = threadIdx.x; e < cnt * k; e += TPB) {
        const int p = e / k;
        const Rel r = rel_pos(coord, idx, (long long)n0 * k + e, n0 + p);
        sPos[e] = make_float4(r.x, r.y, r.z, 0.f);
        sSrc[e] = r.src;
    }
    for (int e = threadIdx.x; e < cnt * k * G; e += TPB) {
        const int r = e / G, g = e - r * G;
        sW[r * G4 + g] = w[(long long)n0 * k * G + e];
    }
    __syncthreads();
    const int I = c / G;
    for (int item = threadIdx.x; item < cnt * c; item += TPB) {
        const int p = item / c, ch = item - p * c;
        const int gl = ch / I;
        const float ax = a[3 * ch], ay = a[3 * ch + 1], az = a[3 * ch + 2], bb = b[ch];
        float accA[G];
#pragma unroll
        for (int g = 0; g < G; ++g) accA[g] = 0.f;
        float ov = 0.f;
        auto slot = [&](int s, float vv) {
            const float4 ps = sPos[p * k + s];
            const float P = pe_act(ax, ay, az, bb, ps.x, ps.y, ps.z);
            const float *wrow = sW + (size_t)(p * k + s) * G4;
            ov = __builtin_fmaf(wrow[gl], vv, ov);
            if (G % 4 == 0) {
#pragma unroll
                for (int g = 0; g < G; g += 4) {
                    const float4 t = *(const float4 *)(wrow + g);
                    accA[g] = __builtin_fmaf(t.x, P, accA[g]);
                    accA[g + 1] = __builtin_fmaf(t.y, P, accA[g + 1]);
                    accA[g + 2] = __builtin_fmaf(t.z, P, accA[g + 2]);
                    accA[g + 3] = __builtin_fmaf(t.w, P, accA[g + 3]);
                }
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) accA[g] = __builtin_fmaf(wrow[g], P, accA[g]);
            }
        };
        if (k == 16) {  // the config's K: all 16 neighbour rows are requested before the first one is consumed
            float vv[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int src = sSrc[p * 16 + s];
                // (G = 6: unconditional, common.h ptv2_zero_pad -- 96 -> 85 us at 120 k points; from G = 12 up the conditional
                // form, whose loads the compiler issues in two waves of eight: 44 vs 56 us at 30 k x 96 with all 16 at once)
                if (G == 6) vv[s] = ptv2_ld_or_zero(v + (long long)src * c + ch, src >= 0);
                else vv[s] = src >= 0 ? v[(long long)src * c + ch] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) slot(s, vv[s]);
        } else {
            for (int s = 0; s < k; ++s) {
                const int src = sSrc[p * k + s];
                slot(s, src >= 0 ? v[(long long)src * c + ch] : 0.f);
            }
        }
        const long long pt = n0 + p;
        out_v[pt * c + ch] = ov;
#pragma unroll
        for (int g = 0; g < G; ++g) A[(pt * G + g) * c + ch] = accA[g];
    }
}

// ================================================================== backward ==
// grad w (pre-mask) and the (ga, gb) partials; one wavefront (64-thread workgroup) per point.
// Three lane mappings, none of which needs a cross-lane reduction inside its inner loop:
//   A  lane = (slot s, group g):      grad w += <g_out[g-th group], v[idx[s], g-th group]>   (I contiguous floats)
//   C  lane = channel:                g_A column in registers; gP[s] = <w[s,:], g_A[:,ch]>; (ga, gb) accumulate in
//                                     registers across slots AND points; the column is also parked in LDS for
//   B  lane = (slot s, channel slice): grad w[s,:] += P[s,ch] g_A[:,ch] over the 64-channel chunk
constexpr int bwd_rounds(int G) { return (8 * G + WAVE - 1) / WAVE; }  // channel rounds for I = C/G <= 8

template <int G>
__global__ __launch_bounds__(WAVE) void aggregate_bwd_tile_kernel(
    int n, int k, int c, const float *__restrict__ w, const float *__restrict__ v, const float *__restrict__ a,
    const float *__restrict__ b, const float *__restrict__ coord, const int *__restrict__ idx,
    const float *__restrict__ g_out, const float *__restrict__ g_A, const float *__restrict__ g_sw,
    float *__restrict__ gw, float *gv_atomic, float *__restrict__ part) {
    extern __shared__ float4 lds4[];
    constexpr int GP = GPof(G), G4 = G4of(G), ROUNDS = bwd_rounds(G);
    const int lane = threadIdx.x;
    float4 *sPos = lds4;                                  // [k]
    float *sW = (float *)(sPos + k);                      // [k][G4]
    float *sGW = sW + (size_t)k * G4;                     // [k][GP]
    float *sT = sGW + (((size_t)k * GP + 3) & ~(size_t)3);  // [64][GP]  g_A chunk
    int *sSrc = (int *)(sT + (((size_t)64 * GP + 3) & ~(size_t)3));  // [k]
    const int I = c / G;
    const int items = G * k;
    const int J = WAVE / k;
    const int ms = lane & (k - 1), mj = lane / k;
    float4 ab[ROUNDS], acc4[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int ch = r * WAVE + lane;
        ab[r] = ch < c ? make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]) : make_float4(0.f, 0.f, 0.f, 0.f);
        acc4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int pt = blockIdx.x; pt < n; pt += gridDim.x) {
        __syncthreads();
        if (lane < k) {
            const Rel r = rel_pos(coord, idx, (long long)pt * k + lane, pt);
            sPos[lane] = make_float4(r.x, r.y, r.z, 0.f);
            sSrc[lane] = r.src;
        }
        for (int item = lane; item < items; item += WAVE) {
            const int s = item / G, g = item - s * G;
            sW[s * G4 + g] = w[(long long)pt * items + item];
        }
        __syncthreads();
        // A: v path
        for (int item = lane; item < items; item += WAVE) {
            const int s = item / G, g = item - s * G;
            const int src = sSrc[s];
            float val = g_sw[(long long)pt * G + g];
            if (src >= 0) {
                const float *go = g_out + (long long)pt * c + g * I, *vr = v + (long long)src * c + g * I;
                for (int i = 0; i < I; ++i) val = __builtin_fmaf(go[i], vr[i], val);
                if (gv_atomic) {
                    const float wv = sW[s * G4 + g];
                    for (int i = 0; i < I; ++i) atomicAdd(gv_atomic + (long long)src * c + g * I + i, go[i] * wv);
                }
            }
            sGW[s * GP + g] = val;
        }
        // C + B per 64-channel chunk
        float accB[G];
#pragma unroll
        for (int g = 0; g < G; ++g) accB[g] = 0.f;
        const float4 psB = sPos[ms];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int cb0 = r * WAVE;
            if (cb0 < c) {  // wave-uniform
                const int ch = cb0 + lane;
                const bool act = ch < c;
                float col[G];
#pragma unroll
                for (int g = 0; g < G; ++g) col[g] = act ? g_A[((long long)pt * G + g) * c + ch] : 0.f;
                // C: lane = channel
                float ga0 = 0.f, ga1 = 0.f, ga2 = 0.f, gb0 = 0.f;
                for (int s = 0; s < k; ++s) {
                    const float4 ps = sPos[s];
                    const float P = pe_act(ab[r].x, ab[r].y, ab[r].z, ab[r].w, ps.x, ps.y, ps.z);
                    const float *wr = sW + s * G4;
                    float gP = 0.f;
#pragma unroll
                    for (int g = 0; g < G; ++g) gP = __builtin_fmaf(wr[g], col[g], gP);
                    const float gpre = P > 0.f ? gP : 0.f;
                    ga0 = __builtin_fmaf(gpre, ps.x, ga0);
                    ga1 = __builtin_fmaf(gpre, ps.y, ga1);
                    ga2 = __builtin_fmaf(gpre, ps.z, ga2);
                    gb0 += gpre;
                }
                acc4[r].x += ga0; acc4[r].y += ga1; acc4[r].z += ga2; acc4[r].w += gb0;
                __syncthreads();  // previous chunk's B readers are done with sT
#pragma unroll
                for (int g = 0; g < G; ++g) sT[lane * GP + g] = col[g];
                __syncthreads();
                // B: lane = (slot ms, slice mj)
                const int cend = (c - cb0) < WAVE ? (c - cb0) : WAVE;
                for (int cl = mj; cl < cend; cl += J) {
                    const int chb = cb0 + cl;
                    const float P = pe_act(a[3 * chb], a[3 * chb + 1], a[3 * chb + 2], b[chb], psB.x, psB.y, psB.z);
                    const float *tr = sT + cl * GP;
#pragma unroll
                    for (int g = 0; g < G; ++g) accB[g] = __builtin_fmaf(P, tr[g], accB[g]);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float t = accB[g];
            for (int o = WAVE >> 1; o >= k; o >>= 1) t += __shfl_xor(t, o, WAVE);
            if (mj == 0) sGW[ms * GP + g] += t;
        }
        __syncthreads();
        for (int item = lane; item < items; item += WAVE) {
            const int s = item / G, g = item - s * G;
            gw[(long long)pt * items + item] = sGW[s * GP + g];
        }
    }
    float4 *mypart = (float4 *)part + (size_t)blockIdx.x * c;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        if (r * WAVE + lane < c) mypart[r * WAVE + lane] = acc4[r];
}

// softmax / Linear(G,G) / ReLU / affine backward per slot; outputs gW1, gz, y; partial [gsc G][gsh G]
template <int G>
__global__ __launch_bounds__(TPB) void aggregate_bwd_rows_kernel(long long rows, int k, const float *__restrict__ W1,
                                                                 const float *__restrict__ sc,
                                                                 const float *__restrict__ sh,
                                                                 const float *__restrict__ Ww2,
                                                                 const float *__restrict__ bw2,
                                                                 const int *__restrict__ idx,
                                                                 const float *__restrict__ gw, float *__restrict__ gW1,
                                                                 float *__restrict__ gz_out, float *__restrict__ y_out,
                                                                 float *__restrict__ part) {
    __shared__ float s_w[WPB][2 * G];
    __shared__ float sWw2[G * G], sBw2[G], sSc[G], sSh[G];
    for (int i = threadIdx.x; i < G * G; i += TPB) sWw2[i] = Ww2[i];
    for (int i = threadIdx.x; i < G; i += TPB) { sBw2[i] = bw2[i]; sSc[i] = sc[i]; sSh[i] = sh[i]; }
    __syncthreads();
    float t_sc[G], t_sh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) t_sc[g] = t_sh[g] = 0.f;
    const long long rows_pad = (rows + WAVE - 1) / WAVE * WAVE;
    for (long long row = (long long)blockIdx.x * TPB + threadIdx.x; row < rows_pad; row += (long long)gridDim.x * TPB) {
        const bool act = row < rows;
        const long long r = act ? row : rows - 1;
        float y[G], gz[G];
        load_row<G>(W1 + r * G, y);
#pragma unroll
        for (int g = 0; g < G; ++g) y[g] = fmaxf(__builtin_fmaf(sSc[g], y[g], sSh[g]), 0.f);
        logits_softmax<G>(y, sWw2, sBw2, k, gz);  // gz holds the unmasked softmax for now
        if (act) store_row<G>(y_out + r * G, y);
        const float valid = (act && idx[r] >= 0) ? 1.f : 0.f;
        {
            float gwr[G];
            load_row<G>(gw + r * G, gwr);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float gm = gwr[g] * valid;
                float dot = gz[g] * gm;
                for (int o = k >> 1; o >= 1; o >>= 1) dot += __shfl_xor(dot, o, WAVE);
                gz[g] = act ? gz[g] * (gm - dot) : 0.f;
            }
        }
        if (act) store_row<G>(gz_out + r * G, gz);
        // Linear(G,G)^T, ReLU mask, BN_w-affine backward, four output channels at a time (keeps only y, gz live)
        constexpr int STEP = (G % 4 == 0) ? 4 : ((G % 2 == 0) ? 2 : 1);
#pragma unroll
        for (int j0 = 0; j0 < G; j0 += STEP) {
            float gu[STEP], u[STEP];
#pragma unroll
            for (int t = 0; t < STEP; ++t) {
                float gy = 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) gy = __builtin_fmaf(gz[g], sWw2[g * G + j0 + t], gy);
                gu[t] = y[j0 + t] > 0.f ? gy : 0.f;
            }
            load_row<STEP>(W1 + r * G + j0, u);
#pragma unroll
            for (int t = 0; t < STEP; ++t) {
                t_sc[j0 + t] = __builtin_fmaf(gu[t], u[t], t_sc[j0 + t]);
                t_sh[j0 + t] += gu[t];
                gu[t] *= sSc[j0 + t];
            }
            if (act) store_row<STEP>(gW1 + r * G + j0, gu);
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float v1 = wave_sum(t_sc[g]), v2 = wave_sum(t_sh[g]);
        if ((threadIdx.x & 63) == 0) { s_w[threadIdx.x >> 6][g] = v1; s_w[threadIdx.x >> 6][G + g] = v2; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * G) {
        float t = 0.f;
        for (int wv = 0; wv < WPB; ++wv) t += s_w[wv][threadIdx.x];
        part[(size_t)blockIdx.x * 2 * G + threadIdx.x] = t;
    }
}

// grad v[j,ch] = sum over slots r that point at j of w[r, g(ch)] * g_out[r / k, ch]
// one thread per (point j, group, float4 of the group's channels): the slot list and the group weight are read
// once per 4 channels instead of once per channel
template <int I>
__global__ __launch_bounds__(TPB) void aggregate_bwd_gv_kernel(int n, int k, int c, int g,
                                                               const float *__restrict__ w,
                                                               const float *__restrict__ g_out,
                                                               const int *__restrict__ inv_ptr,
                                                               const int *__restrict__ inv_rows,
                                                               float *__restrict__ gv, int main_blocks, PtvRiders Rs) {
    if ((int)blockIdx.x >= main_blocks) {  // trailing workgroups: deferred parameter-gradient sums (gva_common.h, riders)
        rider_run(Rs, (int)blockIdx.x - main_blocks);
        return;
    }
    constexpr int V = I >= 4 ? 4 : I;  // channels per thread
    const int cv = c / V;
    const long long total = (long long)n * cv;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)main_blocks * TPB) {
        const int j = (int)(e / cv), ch = (int)(e - (long long)j * cv) * V;
        const int gl = ch / I;
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
        const int p0 = inv_ptr[j], p1 = inv_ptr[j + 1];
        // the list is walked 8 entries at a time: slot ids first, then all weights and gradient rows, then the sums in
        // list order (an entry-by-entry loop pays three dependent memory latencies per entry)
        constexpr int UB = 8;
        // (every load unconditional -- index clamped, -1 or'ed in / zero pad for the entries past the end: common.h,
        // ptv2_zero_pad; as `cond ? load : x` each entry was a basic block of its own, drained one by one)
        int r[UB], rn[UB];
        const int plast = p1 > 0 ? p1 - 1 : 0;
#pragma unroll
        for (int u = 0; u < UB; ++u) r[u] = inv_rows[p0 + u < p1 ? p0 + u : plast] | (p0 + u < p1 ? 0 : -1);
        for (int p = p0; p < p1; p += UB) {
            float wv[UB], t[UB][V];
            // the slot ids of the NEXT batch travel with this batch's weights and gradient rows (one round trip less per
            // batch after the first)
#pragma unroll
            for (int u = 0; u < UB; ++u) rn[u] = inv_rows[p + UB + u < p1 ? p + UB + u : plast] | (p + UB + u < p1 ? 0 : -1);
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const bool ok = r[u] >= 0;
                wv[u] = ptv2_ld_or_zero(w + (long long)r[u] * g + gl, ok);
                const float *go = g_out + (long long)(r[u] / k) * c + ch;
                if (V == 4) {
                    const float4 q = ptv2_ld_or_zero((const float4 *)go, ok);
                    t[u][0] = q.x; t[u][1 % V] = q.y; t[u][2 % V] = q.z; t[u][3 % V] = q.w;
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) t[u][i] = ptv2_ld_or_zero(go + i, ok);
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {  // (entries past the end add w = 0 times 0)
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] = __builtin_fmaf(wv[u], t[u][i], acc[i]);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) r[u] = rn[u];
        }
        if (V == 4) *(float4 *)(gv + (long long)j * c + ch) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else
#pragma unroll
            for (int i = 0; i < V; ++i) gv[(long long)j * c + ch + i] = acc[i];
    }
}

static void launch_bwd_gv(hipStream_t st, int n, int k, int c, int g, const float *w, const float *g_out, const int *inv_ptr,
                          const int *inv_rows, float *gv) {
    const int I = c / g;
    const int V = I >= 4 ? 4 : I;
    const int main_blocks = (int)std::min<long long>(((long long)n * (c / V) + TPB - 1) / TPB, MAX_BLOCKS * 4);
    const PtvRiders Rs = ptv2_rider_take();  // gv depends on none of the parameter-gradient sums queued before this launch
    const dim3 grid((unsigned)(main_blocks + rider_blocks(Rs)));
#define GVCASE(II) case II: hipLaunchKernelGGL(aggregate_bwd_gv_kernel<II>, grid, dim3(TPB), 0, st, n, k, c, g, w, g_out, inv_ptr, inv_rows, gv, main_blocks, Rs); break;
    switch (I) { GVCASE(1) GVCASE(2) GVCASE(4) GVCASE(8) GVCASE(16) GVCASE(32) GVCASE(64) default: break; }
#undef GVCASE
}

struct MapAB {  // columns (ch, j): ga (c,3), gb (c)
    float *ga, *gb;
    __device__ void operator()(int e, double v) const {
        const int ch = e >> 2, j = e & 3;
        if (j < 3) ga[ch * 3 + j] = (float)v; else gb[ch] = (float)v;
    }
};

constexpr int BWD_TILE_BLOCKS = 256 * 16;

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
extern "C" size_t dense_workspace_bytes(int n, int cout, int cin);
extern "C" int linear_wgrad_hip_launcher(int n, int cout, int cin, const float *gY, const float *X, float *dW,
                                         float *db, void *workspace, size_t workspace_bytes, void *stream);

// gva_bwd_point.hip: the fused MFMA backward (one launch) for the (k, c, g) it is instantiated for
int gva_bwd_point_supported(int k, int c, int g);
size_t gva_bwd_point_part_floats(int c, int g);
int gva_bwd_point_launch(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh, const float *Ww2,
                         const float *bw2, const float *v, const float *a, const float *b, const float *coord, const int *idx,
                         const float *g_out, const float *g_A, const float *g_sw, float *gW1, float *gsc, float *gsh,
                         float *gWw2, float *gbw2, float *ga, float *gb, float *part, size_t part_floats_avail,
                         hipStream_t st, const float *Wp2, const float *bp2, PtvDrop drop);
int gva_bwd_point_local(int k, int c, int g);
// gva_bwd_tile.hip: the deep levels' backward per tile of points, g_A formed in the kernel
int gva_bwd_tile_supported(int k, int c, int g);
size_t gva_bwd_tile_part_floats(int n, int c, int g);
int gva_bwd_tile_launch(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh, const float *Ww2,
                        const float *bw2, const float *v, const float *a, const float *b, const float *coord, const int *idx,
                        const float *g_out, const float *Wp2, const float *bp2, float *gW1, float *gsc, float *gsh, float *gWw2,
                        float *gbw2, float *ga, float *gb, float *part, size_t part_floats_avail, gva::PtvDrop drop, hipStream_t st);
// 1 when the backward of this shape runs the tile kernel (AO_AMD_BWD_POINT: the point kernel behind a peb_bwd launch instead)
int gva_bwd_tile_path(int k, int c, int g) {
    return gva_bwd_tile_supported(k, c, g) && !getenv("AO_AMD_BWD_STAGED") && !getenv("AO_AMD_BWD_POINT");
}

int gva_softmax_point_launch(int n, int k, int g, const float *W1, const float *sc, const float *sh, const float *Ww2,
                             const float *bw2, const int *idx, float *w, float *sw, hipStream_t st, PtvDrop drop);

static size_t agg_part_bytes(int n, int k, int c, int g) {
    return align_up(sizeof(float) * std::max({(size_t)BWD_TILE_BLOCKS * 4 * c, (size_t)MAX_BLOCKS * 2 * g, gva_bwd_point_part_floats(c, g),
                                              gva_bwd_tile_supported(k, c, g) ? gva_bwd_tile_part_floats(n, c, g) : (size_t)0}));
}

extern "C" size_t gva_aggregate_workspace_bytes(int n, int k, int c, int g) {
    if (n < 0 || k < 1 || c < 1 || g < 1) return 0;
    const size_t rows = (size_t)n * k;
    const size_t part = agg_part_bytes(n, k, c, g);
    return part + 3 * align_up(sizeof(float) * rows * g) + dense_workspace_bytes((int)std::min<size_t>(rows, 2147483647), g, g) + 1024;
}

extern "C" int gva_aggregate_forward_hip_launcher(int n, int k, int c, int g, const float *W1, const float *sc,
                                                  const float *sh, const float *Ww2, const float *bw2, const float *v,
                                                  const float *a, const float *b, const float *coord, const int *idx,
                                                  float *out_v, float *A, float *sw, float *w, void *stream) {
    if (n < 0 || !pow2(k) || k > 64 || c < 1 || g < 1 || c % g != 0) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)n * k;
    const int nb_rows = (int)std::min<long long>((rows + TPB - 1) / TPB, MAX_BLOCKS * 4);
    {
        PtvScopedTimer t(KID_SOFTMAX_ROWS, st, 4.0 * ((double)rows * (2 * g + 1) + (double)n * g));
        if (k <= 16 && (g == 12 || g == 24 || g == 48 || g == 64) && !getenv("AO_AMD_BWD_STAGED")) {  // g = 6: rows
            const int rc = gva_softmax_point_launch(n, k, g, W1, sc, sh, Ww2, bw2, idx, w, sw, st, ptv2_attn_drop_current());
            if (rc != PTV2_OK) return rc;
        } else {
            const PtvDrop drop = ptv2_attn_drop_current();
#define CALL(GG)                                                                                                                  \
    if (drop.thresh)                                                                                                              \
        hipLaunchKernelGGL((softmax_rows_kernel<GG, true>), dim3(nb_rows), dim3(TPB), 0, st, rows, k, W1, sc, sh, Ww2, bw2, idx, w, \
                           sw, drop);                                                                                             \
    else                                                                                                                          \
        hipLaunchKernelGGL((softmax_rows_kernel<GG, false>), dim3(nb_rows), dim3(TPB), 0, st, rows, k, W1, sc, sh, Ww2, bw2, idx, \
                           w, sw, drop)
            GVA_DISPATCH_G(g, CALL)
#undef CALL
        }
    }
    const int tp = std::max(1, TPB / c);
    const size_t lds = (size_t)tp * k * (sizeof(float4) + sizeof(float) * G4of(g) + sizeof(int));
    {
        // w + idx + coord in; v rows (each unique row once); out_v and A out
        PtvScopedTimer t(KID_AGG_TILE, st, 4.0 * ((double)rows * (g + 1) + (double)n * (3 + 2 * c) + (double)n * g * c));
#define CALL(GG)                                                                                                       \
    hipLaunchKernelGGL(aggregate_tile_kernel<GG>, dim3((n + tp - 1) / tp), dim3(TPB), lds, st, n, k, c, tp, (const float *)w, \
                       v, a, b, coord, idx, out_v, A)
        GVA_DISPATCH_G(g, CALL)
#undef CALL
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// Wp2 / bp2 != NULL: the backward of the grouped projection is done inside the point kernel (g_A, g_sw are not read)
static int aggregate_backward_impl(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh,
                                   const float *Ww2, const float *bw2, const float *v, const float *a, const float *b,
                                   const float *coord, const int *idx, const float *w, const float *g_out, const float *g_A,
                                   const float *g_sw, const float *g_fused_Wp2, const float *g_fused_bp2, const int *inv_ptr,
                                   const int *inv_rows, float *gW1, float *gsc, float *gsh, float *gWw2, float *gbw2, float *gv,
                                   float *ga, float *gb, void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 0 || !pow2(k) || k > 64 || c < 1 || g < 1 || c % g != 0 || !pow2(c / g) || c / g > 64) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < gva_aggregate_workspace_bytes(n, k, c, g)) return PTV2_ERR_WORKSPACE;
    if (n == 0) return PTV2_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)n * k;
    const size_t part_bytes = agg_part_bytes(n, k, c, g);
    const size_t rows_bytes = align_up(sizeof(float) * (size_t)rows * g);
    char *base = (char *)workspace;
    float *part = (float *)base;
    if (inv_ptr && g_fused_Wp2 && g_fused_bp2 && gva_bwd_tile_path(k, c, g)) {
        // the deep levels: one launch per tile of points (gva_bwd_tile.hip), g_A = g_out Wp2 formed per 16-channel chunk in LDS
        {
            // W1, idx, coord, g_out, v rows (unique once) in; gW1 out
            PtvScopedTimer t(KID_BWD_TILE_K + (g == 12 ? 0 : g == 24 ? 1 : g == 48 ? 2 : 3), st,
                             4.0 * ((double)rows * (2 * g + 1) + (double)n * (3 + 2 * c)));
            const PtvDeferScope defer;  // its record sums ride on the gv launch below (which needs none of them)
            const int rc = gva_bwd_tile_launch(n, k, c, g, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, g_fused_Wp2, g_fused_bp2, gW1,
                                               gsc, gsh, gWw2, gbw2, ga, gb, part, part_bytes / sizeof(float), ptv2_attn_drop_current(), st);
            if (rc != PTV2_OK) return rc;
        }
        {
            PtvScopedTimer t(KID_BWD_GV, st, 4.0 * ((double)rows * (g + 1) + 2.0 * n * c + n));
            launch_bwd_gv(st, n, k, c, g, w, g_out, inv_ptr, inv_rows, gv);
        }
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    if (inv_ptr && gva_bwd_point_supported(k, c, g) && !getenv("AO_AMD_BWD_STAGED")) {
        // one fused MFMA launch (+ its finalize) instead of tile / rows / finalizes / the G x G weight-gradient GEMM
        {
            // W1, idx, coord, g_out, g_sw, v rows (unique once), g_A in; gW1 out
            PtvScopedTimer t(KID_BWD_POINT + (g == 6 ? 0 : g == 12 ? 1 : g == 24 ? 2 : g == 48 ? 3 : 4), st, 4.0 * ((double)rows * (2 * g + 1) + (double)n * (3 + 2 * c + g) + (double)n * g * c));
            const PtvDeferScope defer;  // its record sums ride on the gv launch below (which needs none of them)
            const int rc = gva_bwd_point_launch(n, k, c, g, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, g_A, g_sw, gW1, gsc,
                                                gsh, gWw2, gbw2, ga, gb, part, part_bytes / sizeof(float), st, g_fused_Wp2,
                                                g_fused_bp2, ptv2_attn_drop_current());
            if (rc != PTV2_OK) return rc;
        }
        {
            PtvScopedTimer t(KID_BWD_GV, st, 4.0 * ((double)rows * (g + 1) + 2.0 * n * c + n));
            launch_bwd_gv(st, n, k, c, g, w, g_out, inv_ptr, inv_rows, gv);
        }
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    if (ptv2_attn_drop_current().thresh) return PTV2_ERR_ARG;  // attention dropout: the fused point kernel only
    float *gw = (float *)(base + part_bytes);
    float *gz = (float *)(base + part_bytes + rows_bytes);
    float *yb = (float *)(base + part_bytes + 2 * rows_bytes);
    void *dense_ws = base + part_bytes + 3 * rows_bytes;
    const size_t dense_bytes = workspace_bytes - (part_bytes + 3 * rows_bytes);

    constexpr int dummy = 0;
    (void)dummy;
    const size_t lds_tile = sizeof(float4) * (size_t)k +
                            sizeof(float) * ((size_t)k * G4of(g) + (((size_t)k * GPof(g) + 3) & ~(size_t)3) +
                                             (((size_t)64 * GPof(g) + 3) & ~(size_t)3)) + sizeof(int) * k;
    if (lds_tile > 160 * 1024 || c > WAVE * bwd_rounds(g)) return PTV2_ERR_ARG;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / lds_tile));
    const int nb_tile = std::min(n, std::min(256 * per_cu, (int)BWD_TILE_BLOCKS));
#define CALL(GG)                                                                                                        \
    if (lds_tile > 32 * 1024)                                                                                           \
        (void)hipFuncSetAttribute((const void *)aggregate_bwd_tile_kernel<GG>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds_tile);                                                                       \
    hipLaunchKernelGGL(aggregate_bwd_tile_kernel<GG>, dim3(nb_tile), dim3(WAVE), lds_tile, st, n, k, c, w, v, a, b, coord, idx, \
                       g_out, g_A, g_sw, gw, inv_ptr ? (float *)nullptr : gv, part)
    {
        // w, idx, coord, g_out, g_sw, v rows (unique once), g_A in; grad w out
        PtvScopedTimer t(KID_BWD_TILE, st, 4.0 * ((double)rows * (2 * g + 1) + (double)n * (3 + 2 * c + g) + (double)n * g * c));
        GVA_DISPATCH_G(g, CALL)
    }
#undef CALL
    launch_finalize(st, (const float *)part, nb_tile, 4 * c, MapAB{ga, gb});
    if (inv_ptr)
    {
        PtvScopedTimer t(KID_BWD_GV, st, 4.0 * ((double)rows * (g + 1) + 2.0 * n * c + n));
        launch_bwd_gv(st, n, k, c, g, w, g_out, inv_ptr, inv_rows, gv);
    }
    const int nb_rows = (int)std::min<long long>((rows + TPB - 1) / TPB, MAX_BLOCKS);
#define CALL(GG)                                                                                                      \
    hipLaunchKernelGGL(aggregate_bwd_rows_kernel<GG>, dim3(nb_rows), dim3(TPB), 0, st, rows, k, W1, sc, sh, Ww2, bw2, idx, \
                       (const float *)gw, gW1, gz, yb, part)
    {
        PtvScopedTimer t(KID_BWD_ROWS, st, 4.0 * (double)rows * (5 * g + 1));
        GVA_DISPATCH_G(g, CALL)
    }
#undef CALL
    launch_finalize(st, (const float *)part, nb_rows, 2 * g, MapSplit2<float>{gsc, gsh, g});
    PTV2_CHECK_LAUNCH();
    // grad Ww2[g][g'] = sum_rows gz[r,g] y[r,g'],  grad bw2 = column sums of gz: the Linear weight-gradient reduction
    return linear_wgrad_hip_launcher((int)rows, g, g, gz, yb, gWw2, gbw2, dense_ws, dense_bytes, stream);
}

// gva_aggregate_backward with the backward of the grouped projection (gva_peb_backward) folded into the point kernel:
// for the instances gva_bwd_point_local() names, g_A (N,G,C) and g_sw are never materialised.  Internal to block runtime.
int gva_aggregate_backward_fused_peb(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh,
                                     const float *Ww2, const float *bw2, const float *v, const float *a, const float *b,
                                     const float *coord, const int *idx, const float *w, const float *g_out, const float *Wp2,
                                     const float *bp2, const int *inv_ptr, const int *inv_rows, float *gW1, float *gsc,
                                     float *gsh, float *gWw2, float *gbw2, float *gv, float *ga, float *gb, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    if (!inv_ptr || !(gva_bwd_point_local(k, c, g) || gva_bwd_tile_path(k, c, g)) || !Wp2 || !bp2) return PTV2_ERR_ARG;
    return aggregate_backward_impl(n, k, c, g, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, w, g_out, nullptr, nullptr, Wp2, bp2,
                                   inv_ptr, inv_rows, gW1, gsc, gsh, gWw2, gbw2, gv, ga, gb, workspace, workspace_bytes, stream);
}

// the public form of the above: backward of gva_attention_forward_hip_launcher (the grouped projection's backward folded in)
extern "C" int gva_attention_backward_hip_launcher(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh,
                                                   const float *Ww2, const float *bw2, const float *v, const float *a,
                                                   const float *b, const float *coord, const int *idx, const float *w,
                                                   const float *g_out, const float *Wp2, const float *bp2, const int *inv_ptr,
                                                   const int *inv_rows, float *gW1, float *gsc, float *gsh, float *gWw2, float *gbw2,
                                                   float *gv, float *ga, float *gb, void *workspace, size_t workspace_bytes,
                                                   void *stream) {
    if (!gva_bwd_tile_supported(k, c, g)) return PTV2_ERR_ARG;
    return gva_aggregate_backward_fused_peb(n, k, c, g, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, w, g_out, Wp2, bp2, inv_ptr, inv_rows,
                                            gW1, gsc, gsh, gWw2, gbw2, gv, ga, gb, workspace, workspace_bytes, stream);
}

extern "C" int gva_aggregate_backward_hip_launcher(int n, int k, int c, int g, const float *W1, const float *sc,
                                                   const float *sh, const float *Ww2, const float *bw2,
                                                   const float *v, const float *a, const float *b,
                                                   const float *coord, const int *idx, const float *w,
                                                   const float *g_out, const float *g_A, const float *g_sw,
                                                   const int *inv_ptr, const int *inv_rows, float *gW1, float *gsc,
                                                   float *gsh, float *gWw2, float *gbw2, float *gv, float *ga,
                                                   float *gb, void *workspace, size_t workspace_bytes, void *stream) {
    if (!g_A || !g_sw) return PTV2_ERR_ARG;
    return aggregate_backward_impl(n, k, c, g, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, w, g_out, g_A, g_sw, nullptr, nullptr,
                                   inv_ptr, inv_rows, gW1, gsc, gsh, gWw2, gbw2, gv, ga, gb, workspace, workspace_bytes, stream);
}
