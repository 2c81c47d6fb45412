// ao_amd/csrc/gva_bwd_tile.hip -- backward of the softmax / aggregation / grouped-projection stages of grouped vector attention
// at the DEEP levels ((G, C) = (12, 96), (24, 192), (48, 384), (64, 512); K = 16) as ONE launch: the backward of the grouped
// projection (g_A = g_out Wp2 per group, an (N, G, C) tensor that peb_bwd_kernel wrote and attention_bwd_point_kernel read
// twice) is formed inside the kernel, per tile of points and 16-channel chunk, and lives in LDS only.
//
// Reference op: the backward of GroupedVectorAttention.forward, point_transformer_v2m2_base.py:103-129 (autograd through the
// softmax, the einsum "n s g i, n s g -> n g i" and linear_p_bias / weight_encoding), re-associated as ao_amd/ptv2/gva.py
// describes.  Per point (s = slot, g / j = group, c' = channel of the positional encoding, o = output channel):
//   g_A (g,c')  = sum_i g_out[8 g + i] Wp2[8 g + i, c']        g_sw[g] = sum_i g_out[8 g + i] bp2[8 g + i]
//   gw  (s,g)   = sum_c' g_A[g,c'] P[s,c'] + <g_out, v[idx[s]]>_g + g_sw[g]                 P = ReLU(a . pos + b)
//   gP  (s,c')  = sum_g w[s,g] g_A[g,c']   -> (ga, gb)[c'] += relu'(P) gP (pos, 1)
//   gz = sm (gm - <sm, gm>_s),  gy = gz Ww2,  gW1 = relu'(y) gy sc,  (gsc, gsh, gWw2, gbw2) as sums over all slots
// g_A needs Wp2 (C x C) once per point unless points share it, so a workgroup owns a TILE of TP points (a wavefront TP / 4 of
// them, whole: no cross-wave sum of gw) and walks the channels in chunks of 16:
//   a   g_A chunk of the tile on the matrix cores (rows = points, 2 instructions per group; the wavefronts split the groups;
//       the Wp2 piece of the next chunk is in flight) -> LDS, double-buffered
//   b   per point: gw^T += g_A P^T (the chunk's 16 channels = 4 contraction steps) and gP = w g_A with the (ga, gb) sums of
//       the chunk's channels -- both read the point's g_A rows from LDS in their operand layouts
// with the softmax re-evaluation in front and the softmax / Linear(G,G) backward behind, as in attention_bwd_point_kernel
// (gva_bwd_point.hip), whose record layout [4C (ga.xyz, gb)] [G gsc] [G gsh] [G*G gWw2] [G gbw2] and finalize it shares.
// One workgroup barrier per chunk; no float atomics; every sum in a fixed order.
#include <algorithm>
#include <cstdlib>

#include "gva_common.h"

namespace gva {

typedef float bt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bt_v4f bt_mfma(float a, float b, bt_v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <int CTRL>
__device__ __forceinline__ float bt_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float bt_row16_sum(float v) {
    v += bt_dpp<0xB1>(v);
    v += bt_dpp<0x4E>(v);
    v += bt_dpp<0x141>(v);
    v += bt_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float bt_row16_max(float v) {
    v = fmaxf(v, bt_dpp<0xB1>(v));
    v = fmaxf(v, bt_dpp<0x4E>(v));
    v = fmaxf(v, bt_dpp<0x141>(v));
    v = fmaxf(v, bt_dpp<0x140>(v));
    return v;
}
// LDS hand-off inside a wavefront: its LDS operations complete in order once counted down; the compiler must not move LDS
// accesses across.  (A wavefront-scope fence also waits for the wavefront's GLOBAL stores -- vmcnt(0): ~2 us per point behind
// the gW1 stores of the tail.)
__device__ __forceinline__ void bt_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// v + the same lane of the other quarters (lane ^ 16, then lane ^ 32) without the LDS crossbar: gfx950 swaps rows / halves
__device__ __forceinline__ float bt_quarter_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float h = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(h), __float_as_uint(h), false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}
__host__ __device__ constexpr int bt_ww_pitch(int G) {  // >= G, = 4 mod 16
    int p = G;
    while (p % 16 != 4) ++p;
    return p;
}
__host__ __device__ constexpr int bt_point_pitch(int G, int GP) {  // >= G GP, = 4 mod 64
    int p = G * GP;
    while (p % 64 != 4) p += 4;
    return p;
}

template <int G, int C, int TP>
struct BwdTileCfg {
    static constexpr int GT = (G + 15) / 16, G16 = GT * 16, GPW = bt_ww_pitch(G), PPW = TP / 4, NCH = C / 16, NGW = G / 4, GP = 20,
                         PPG = bt_point_pitch(G, GP), PF = 4 * C + 3 * G + G * G,
                         DW = 2 * G16 * 17 > 3 * G + G * G ? 2 * G16 * 17 : 3 * G + G * G;  // a wavefront's transposes, then its record piece
    static_assert((TP == 4 || TP == 8 || TP == 16) && G % 4 == 0 && C == 8 * G && GPW >= G16, "tiles of 8 or 16 points");
    // [sAB 4C] [sWw G16 GPW] [sBw, sSc, sSh 3 G16] [sPos 4 TP 16] [sGo TP C] [sGsw TP G16] [sFinAB 4 x 4C] [sGA 2 TP PPG | sT 4 DW]
    static constexpr size_t tail = std::max<size_t>(2 * (size_t)TP * PPG, 4 * (size_t)DW);
    static constexpr size_t lds_floats = 4 * (size_t)C + (size_t)G16 * GPW + 3 * G16 + 4 * TP * 16 + (size_t)TP * C + TP * G16 + 16 * (size_t)C + tail;
};

// timing-only builds (-DBT_STAMPS): wavefront 0 of workgroups 0 and gridDim.x / 2 leaves the 100 MHz wall clock at the phase
// boundaries behind the records (tools/bench_bwd_tile.py --stamps reads them)
#ifdef BT_STAMPS
#define BT_STAMP(k) do { if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) \
    ((unsigned long long *)(part + (size_t)gridDim.x * PF + 64))[(blockIdx.x ? 16 : 0) + (k)] = wall_clock64(); } while (0)
#else
#define BT_STAMP(k)
#endif

template <int G, int C, int TP, bool DROP>
__device__ __forceinline__ void attention_bwd_tile_body(
    int n, const float *__restrict__ W1, const float *__restrict__ sc, const float *__restrict__ sh, const float *__restrict__ Ww2,
    const float *__restrict__ bw2, const float *__restrict__ v, const float *__restrict__ a, const float *__restrict__ b,
    const float *__restrict__ coord, const int *__restrict__ idx, const float *__restrict__ g_out, const float *__restrict__ Wp2,
    const float *__restrict__ bp2, float *__restrict__ gW1, float *__restrict__ part, PtvDrop drop, const long long pt0) {
    using K = BwdTileCfg<G, C, TP>;
    constexpr int GT = K::GT, G16 = K::G16, GPW = K::GPW, PPW = K::PPW, NCH = K::NCH, NGW = K::NGW, GP = K::GP, PPG = K::PPG, PF = K::PF,
                  DW = K::DW;
    extern __shared__ float4 lds4[];
    float4 *sAB = lds4;                              // [C]  (a.xyz, b)
    float4 *sPos = sAB + C;                          // [TP][16]
    float4 *sFinAB = sPos + TP * 16;                 // [4 waves][C]  (ga.xyz, gb) of a wavefront's points
    float *sWw = (float *)(sFinAB + 4 * C);          // [G16][GPW]  Ww2, zero padded
    float *sBw = sWw + G16 * GPW;                    // [G16]
    float *sSc = sBw + G16;
    float *sSh = sSc + G16;
    float *sGo = sSh + G16;                          // [TP][C]   g_out rows of the tile
    float *sGsw = sGo + TP * C;                      // [TP][G16]
    float *sGA = sGsw + TP * G16;                    // [2][TP][PPG]  g_A chunk (point; group, 16 c' + pad)
    float *sT = sGA;                                 // behind the chunk loop: [4 waves][2][G16][17] transposes of gz, y
    //                                                                   then [4 waves][3 G + G G] the wavefronts' record pieces

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const long long last = (long long)n - 1;  // (pt0: the first point of this workgroup's tile)
    BT_STAMP(0);

    // ---- requests first: neighbour ids, logits rows, g_out rows of my points
    long long pts[PPW];
    bool act[PPW];
    int srcv[PPW];
    float u1[PPW][GT][4];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const long long pt = pt0 + PPW * wid + i;
        act[i] = pt < n;
        pts[i] = act[i] ? pt : last;
        srcv[i] = idx[pts[i] * 16 + l15];
        const float *row = W1 + (pts[i] * 16 + l15) * G;
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int j0 = 16 * t + 4 * q;
            const float4 uu = *(const float4 *)(row + (j0 < G ? j0 : 0));
            u1[i][t][0] = j0 < G ? uu.x : 0.f; u1[i][t][1] = j0 < G ? uu.y : 0.f;
            u1[i][t][2] = j0 < G ? uu.z : 0.f; u1[i][t][3] = j0 < G ? uu.w : 0.f;
        }
        for (int e = lane; e < C; e += WAVE) sGo[(PPW * wid + i) * C + e] = act[i] ? g_out[pts[i] * C + e] : 0.f;
    }
    for (int ch = tid; ch < C; ch += 256) sAB[ch] = make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]);
    for (int e = tid; e < G16 * GPW; e += 256) {
        const int g = e / GPW, j = e - g * GPW;
        sWw[e] = (g < G && j < G) ? Ww2[g * G + j] : 0.f;
    }
    for (int g = tid; g < G16; g += 256) {
        sBw[g] = g < G ? bw2[g] : 0.f;
        sSc[g] = g < G ? sc[g] : 0.f;
        sSh[g] = g < G ? sh[g] : 0.f;
    }
    float3 myp[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const long long ss = srcv[i] >= 0 ? srcv[i] : 0;
        const float x = coord[3 * ss] - coord[3 * pts[i]], y = coord[3 * ss + 1] - coord[3 * pts[i] + 1], z = coord[3 * ss + 2] - coord[3 * pts[i] + 2];
        myp[i] = srcv[i] >= 0 ? make_float3(x, y, z) : make_float3(0.f, 0.f, 0.f);
        sPos[(PPW * wid + i) * 16 + l15] = make_float4(myp[i].x, myp[i].y, myp[i].z, 0.f);
    }
    BT_STAMP(1);
    __syncthreads();
    BT_STAMP(2);

    // ---- softmax re-evaluation: y = ReLU(sc W1 + sh) in the layout lane = (s = l15; j = 16 t + 4 q + r); z^T = Ww2 y^T + bw2
    float sm[PPW][GT][4];   // the unmasked softmax (rows g = 16 t + 4 q + r, column s = l15)
    float dr[DROP ? PPW : 1][GT][4];  // attention dropout factors
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        float y[GT][4];
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const float4 s4 = *(const float4 *)(sSc + 16 * t + 4 * q), h4 = *(const float4 *)(sSh + 16 * t + 4 * q);
            y[t][0] = fmaxf(__builtin_fmaf(s4.x, u1[i][t][0], h4.x), 0.f);
            y[t][1] = fmaxf(__builtin_fmaf(s4.y, u1[i][t][1], h4.y), 0.f);
            y[t][2] = fmaxf(__builtin_fmaf(s4.z, u1[i][t][2], h4.z), 0.f);
            y[t][3] = fmaxf(__builtin_fmaf(s4.w, u1[i][t][3], h4.w), 0.f);
        }
#pragma unroll
        for (int tg = 0; tg < GT; ++tg) {
            bt_v4f z = (bt_v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const float4 w4 = *(const float4 *)(sWw + (16 * tg + l15) * GPW + 16 * t + 4 * q);
                z = bt_mfma(w4.x, y[t][0], z);
                z = bt_mfma(w4.y, y[t][1], z);
                z = bt_mfma(w4.z, y[t][2], z);
                z = bt_mfma(w4.w, y[t][3], z);
            }
            const float4 b4 = *(const float4 *)(sBw + 16 * tg + 4 * q);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float zz = z[r] + bb[r];
                const float mx = bt_row16_max(zz);
                // (hardware exp2 / reciprocal, ~1e-6 relative, as attention_bwd_point_kernel re-evaluates it)
                const float e = __builtin_amdgcn_exp2f((zz - mx) * 1.44269504088896340736f);
                const float den = bt_row16_sum(e);
                sm[i][tg][r] = (16 * tg + 4 * q + r < G) ? e * __builtin_amdgcn_rcpf(den) : 0.f;
                if (DROP) dr[i][tg][r] = ptv2_drop_factor(drop, ((unsigned long long)pts[i] * 16 + l15) * G + ((16 * tg + 4 * q + r) & 0xffff));
            }
        }
    }
    BT_STAMP(3);
    // ---- g_sw of my points (their g_out rows are wave-private so far)
    bt_wave_sync();
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = PPW * wid + i;
        if (lane < G16) {
            float acc = 0.f;
            if (lane < G) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(sGo[p * C + 8 * lane + e], bp2[8 * lane + e], acc);
            }
            sGsw[p * G16 + lane] = acc;
        }
    }
    // ---- gw^T starts from the v path: <g_out, v[idx[s]]> over the 8 channels of group g = 16 t + 4 q + r, slot s = l15
    bt_v4f gwT[PPW][GT];
#ifdef BT_SKIP_VPATH
#pragma unroll
    for (int i = 0; i < PPW; ++i)
#pragma unroll
        for (int t = 0; t < GT; ++t) gwT[i][t] = (bt_v4f){0.f, 0.f, 0.f, 0.f};
#else
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = PPW * wid + i;
        const bool ok = srcv[i] >= 0 && act[i];
        const float *vrow = ok ? v + (long long)srcv[i] * C : ptv2_zero_pad;
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            float4 v0[4], v1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = 16 * t + 4 * q + r;
                const float *vp = (ok && g < G) ? vrow + 8 * g : ptv2_zero_pad;
                v0[r] = *(const float4 *)vp;
                v1[r] = *(const float4 *)(vp + 4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = 16 * t + 4 * q + r;
                const float *gop = sGo + p * C + 8 * (g < G ? g : 0);
                const float4 g0 = *(const float4 *)gop, g1 = *(const float4 *)(gop + 4);
                float acc = g0.x * v0[r].x;
                acc = __builtin_fmaf(g0.y, v0[r].y, acc); acc = __builtin_fmaf(g0.z, v0[r].z, acc); acc = __builtin_fmaf(g0.w, v0[r].w, acc);
                acc = __builtin_fmaf(g1.x, v1[r].x, acc); acc = __builtin_fmaf(g1.y, v1[r].y, acc);
                acc = __builtin_fmaf(g1.z, v1[r].z, acc); acc = __builtin_fmaf(g1.w, v1[r].w, acc);
                gwT[i][t][r] = acc;
            }
        }
    }
#endif
    BT_STAMP(4);
    __syncthreads();  // every point's g_out row is in LDS
    BT_STAMP(5);

    // ---- the chunks of 16 channels c'
    float3 rp[PPW][4];     // relative positions of the slots 4 q + r (the rows of the gP tiles): loop-invariant
#pragma unroll
    for (int i = 0; i < PPW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 t = sPos[(PPW * wid + i) * 16 + 4 * q + r];
            rp[i][r] = make_float3(t.x, t.y, t.z);
        }
    float goA[NGW][2];     // A operand of the g_A product: g_out[point l15, 8 g + 4 ks + q]
    const float *wprow[NGW][2];
    float wpn[NGW][2];
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int o = 8 * (wid + 4 * gi) + 4 * ks + q;
            goA[gi][ks] = sGo[(l15 < TP ? l15 : TP - 1) * C + o];
            wprow[gi][ks] = Wp2 + (size_t)o * C + l15;
            wpn[gi][ks] = *wprow[gi][ks];
        }
#ifdef BT_SKIP_CHUNKS
    for (int e = tid; e < 4 * C; e += 256) sFinAB[e] = make_float4(0.f, 0.f, 0.f, 0.f);
#else
#pragma unroll 2
    for (int ck = 0; ck < NCH; ++ck) {
        float *buf = sGA + (ck & 1) * TP * PPG;
        // a: g_A chunk of the tile, my groups
#ifndef BT_NO_A
        {
            float wpc[NGW][2];
            const int cn = ck + 1 < NCH ? ck + 1 : ck;
#pragma unroll
            for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) { wpc[gi][ks] = wpn[gi][ks]; wpn[gi][ks] = wprow[gi][ks][16 * cn]; }
            bt_v4f da[NGW];
#pragma unroll
            for (int gi = 0; gi < NGW; ++gi) da[gi] = bt_mfma(goA[gi][0], wpc[gi][0], (bt_v4f){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int gi = 0; gi < NGW; ++gi) da[gi] = bt_mfma(goA[gi][1], wpc[gi][1], da[gi]);
            if (4 * q < TP) {
#pragma unroll
                for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) buf[(4 * q + r) * PPG + (wid + 4 * gi) * GP + l15] = da[gi][r];
            }
        }
#endif
        __syncthreads();
        // b: my points
        float4 abq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) abq[e] = sAB[16 * ck + 4 * q + e];
        const float4 abl = sAB[16 * ck + l15];
        float4 accab = make_float4(0.f, 0.f, 0.f, 0.f);
        // operands of both products for all my points first, then the matrix instructions with the points' chains interleaved
        // (a chain of dependent instructions runs at 40 cycles per step against 32 issued)
        float Pq[PPW][4];
        float4 g4[PPW][GT];
        float bv[PPW][GT][4];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const float *ga = buf + (PPW * wid + i) * PPG;
#pragma unroll
            for (int e = 0; e < 4; ++e) Pq[i][e] = pe_act(abq[e].x, abq[e].y, abq[e].z, abq[e].w, myp[i].x, myp[i].y, myp[i].z);
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const int g = 16 * t + l15;
                g4[i][t] = *(const float4 *)(ga + (g < G ? g : G - 1) * GP + 4 * q);
                if (g >= G) g4[i][t] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gr = 16 * t + 4 * q + r;
                    bv[i][t][r] = (16 * t + r < G) ? ga[(gr < G ? gr : G - 1) * GP + l15] : 0.f;
                }
            }
        }
        // gw^T (g, s) += g_A (g, c') P^T (c', s): contraction step e of quarter q is channel 16 ck + 4 q + e
#ifndef BT_NO_B3
#pragma unroll
        for (int t = 0; t < GT; ++t) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) gwT[i][t] = bt_mfma(g4[i][t].x, Pq[i][0], gwT[i][t]);
#pragma unroll
            for (int i = 0; i < PPW; ++i) gwT[i][t] = bt_mfma(g4[i][t].y, Pq[i][1], gwT[i][t]);
#pragma unroll
            for (int i = 0; i < PPW; ++i) gwT[i][t] = bt_mfma(g4[i][t].z, Pq[i][2], gwT[i][t]);
#pragma unroll
            for (int i = 0; i < PPW; ++i) gwT[i][t] = bt_mfma(g4[i][t].w, Pq[i][3], gwT[i][t]);
        }
#endif
        // gP (s, c') = w g_A: rows s = 4 q + r, column c' = 16 ck + l15; contraction over the groups, two accumulators per point
        bt_v4f d0[PPW], d1[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) d0[i] = d1[i] = (bt_v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#ifndef BT_NO_B4
                if (16 * t + r < G)
#else
                if (16 * t + r < 1)
#endif
                {  // (compile-time: a step whose four rows are all padding is skipped; padding rows carry w = 0)
#pragma unroll
                    for (int i = 0; i < PPW; ++i) {
                        const bool valid = srcv[i] >= 0 && act[i];
                        const float wm = valid ? (DROP ? sm[i][t][r] * dr[DROP ? i : 0][t][r] : sm[i][t][r]) : 0.f;
                        if (r & 1) d1[i] = bt_mfma(wm, bv[i][t][r], d1[i]);
                        else d0[i] = bt_mfma(wm, bv[i][t][r], d0[i]);
                    }
                }
            }
#ifdef BT_NO_EPI
#pragma unroll
        for (int i = 0; i < PPW; ++i) accab.x += d0[i][0] + d1[i][1];
        if (accab.x == 123.f) sFinAB[wid * C + 16 * ck + l15] = accab;
#else
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float P = pe_act(abl.x, abl.y, abl.z, abl.w, rp[i][r].x, rp[i][r].y, rp[i][r].z);
                const float gpre = P > 0.f ? d0[i][r] + d1[i][r] : 0.f;
                accab.x = __builtin_fmaf(gpre, rp[i][r].x, accab.x);
                accab.y = __builtin_fmaf(gpre, rp[i][r].y, accab.y);
                accab.z = __builtin_fmaf(gpre, rp[i][r].z, accab.z);
                accab.w += gpre;
            }
        }
        accab.x = bt_quarter_sum(accab.x); accab.y = bt_quarter_sum(accab.y);
        accab.z = bt_quarter_sum(accab.z); accab.w = bt_quarter_sum(accab.w);
        if (q == 0) sFinAB[wid * C + 16 * ck + l15] = accab;
#endif
    }
#endif
    BT_STAMP(6);
    __syncthreads();  // the g_A buffers are free: transposes and the record take their place
    BT_STAMP(7);

    // ---- softmax backward, Linear(G,G) backward, the sums over all slots
    float *mGz = sT + wid * DW, *mY = mGz + G16 * 17;
    float tsc[GT][4], tsh[GT][4], gbw[GT][4];
    bt_v4f accW[GT * GT];
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) tsc[t][r] = tsh[t][r] = gbw[t][r] = 0.f;
#pragma unroll
    for (int e = 0; e < GT * GT; ++e) accW[e] = (bt_v4f){0.f, 0.f, 0.f, 0.f};
#ifndef BT_SKIP_TAIL
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = PPW * wid + i;
        const bool valid = srcv[i] >= 0 && act[i];
        float y[GT][4], gz[GT][4];
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const float4 s4 = *(const float4 *)(sSc + 16 * t + 4 * q), h4 = *(const float4 *)(sSh + 16 * t + 4 * q);
            y[t][0] = fmaxf(__builtin_fmaf(s4.x, u1[i][t][0], h4.x), 0.f);
            y[t][1] = fmaxf(__builtin_fmaf(s4.y, u1[i][t][1], h4.y), 0.f);
            y[t][2] = fmaxf(__builtin_fmaf(s4.z, u1[i][t][2], h4.z), 0.f);
            y[t][3] = fmaxf(__builtin_fmaf(s4.w, u1[i][t][3], h4.w), 0.f);
            const float4 gs4 = *(const float4 *)(sGsw + p * G16 + 16 * t + 4 * q);
            const float gs[4] = {gs4.x, gs4.y, gs4.z, gs4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gm = valid ? gwT[i][t][r] + gs[r] : 0.f;
                if (DROP) gm *= dr[DROP ? i : 0][t][r];  // (the weight that reached the aggregation was sm * D)
                const float dot = bt_row16_sum(sm[i][t][r] * gm);
                gz[t][r] = act[i] ? sm[i][t][r] * (gm - dot) : 0.f;
            }
        }
        bt_wave_sync();  // (the previous point's transposes have been read)
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mGz[(16 * t + 4 * q + r) * 17 + l15] = gz[t][r];
                mY[(16 * t + 4 * q + r) * 17 + l15] = y[t][r];
                gbw[t][r] += gz[t][r];
            }
        // gy^T (j, s) = Ww2^T gz^T -> gW1, gsc, gsh
#pragma unroll
        for (int tj = 0; tj < GT; ++tj) {
            bt_v4f gy = (bt_v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tg = 0; tg < GT; ++tg)
#pragma unroll
                for (int r = 0; r < 4; ++r) gy = bt_mfma(sWw[(16 * tg + 4 * q + r) * GPW + 16 * tj + l15], gz[tg][r], gy);
            const float4 s4 = *(const float4 *)(sSc + 16 * tj + 4 * q);
            const float scv[4] = {s4.x, s4.y, s4.z, s4.w};
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gu = y[tj][r] > 0.f ? gy[r] : 0.f;
                tsc[tj][r] = __builtin_fmaf(gu, u1[i][tj][r], tsc[tj][r]);
                tsh[tj][r] += gu;
                o[r] = gu * scv[r];
            }
            if (act[i] && 16 * tj + 4 * q < G) *(float4 *)(gW1 + (pts[i] * 16 + l15) * G + 16 * tj + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
        }
        // gWw2 (g, j) += gz^T y: contraction over the slots, from the transposed tiles
        bt_wave_sync();
#pragma unroll
        for (int tg = 0; tg < GT; ++tg)
#pragma unroll
            for (int tj = 0; tj < GT; ++tj)
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    accW[tg * GT + tj] = bt_mfma(mGz[(16 * tg + l15) * 17 + 4 * st + q], mY[(16 * tj + l15) * 17 + 4 * st + q], accW[tg * GT + tj]);
    }
#else
    if (wid == 99) gW1[0] = gwT[0][0][0] + sm[0][0][0] + u1[0][0][0] + mGz[0] + mY[0];
#endif

    BT_STAMP(8);
    // ---- the workgroup's record: each wavefront leaves its sums in its own piece of LDS; one barrier; every thread adds the four
    // pieces of its elements in wavefront order ((w0 + w1) + w2) + w3 and stores them
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { tsc[t][r] = bt_row16_sum(tsc[t][r]); tsh[t][r] = bt_row16_sum(tsh[t][r]); gbw[t][r] = bt_row16_sum(gbw[t][r]); }
    bt_wave_sync();  // (the last point's transposes have been read: the piece is this wavefront's own)
    {
        float *mine = sT + wid * DW;  // [gsc G][gsh G][gbw2 G][gWw2 G x G]
        if (l15 == 0) {
#pragma unroll
            for (int t = 0; t < GT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * t + 4 * q + r;
                    if (j < G) { mine[j] = tsc[t][r]; mine[G + j] = tsh[t][r]; mine[2 * G + j] = gbw[t][r]; }
                }
        }
#pragma unroll
        for (int tg = 0; tg < GT; ++tg)
#pragma unroll
            for (int tj = 0; tj < GT; ++tj) {
                const int j = 16 * tj + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = 16 * tg + 4 * q + r;
                    if (g < G && j < G) mine[3 * G + g * G + j] = accW[tg * GT + tj][r];
                }
            }
    }
    __syncthreads();
    float *rec = part + (size_t)blockIdx.x * PF;
    for (int ch = tid; ch < C; ch += 256) {
        const float4 a0 = sFinAB[ch], a1 = sFinAB[C + ch], a2 = sFinAB[2 * C + ch], a3 = sFinAB[3 * C + ch];
        *(float4 *)(rec + 4 * ch) = make_float4(((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y,
                                                ((a0.z + a1.z) + a2.z) + a3.z, ((a0.w + a1.w) + a2.w) + a3.w);
    }
    for (int e = tid; e < 3 * G + G * G; e += 256) {
        const float s = ((sT[e] + sT[DW + e]) + sT[2 * DW + e]) + sT[3 * DW + e];
        // record layout (MapBwdPoint): [ga, gb 4C][gsc G][gsh G][gWw2 G x G][gbw2 G]
        const int at = e < 2 * G ? e : e < 3 * G ? e + G * G : e - G;
        rec[4 * C + at] = s;
    }
    BT_STAMP(9);
}

template <int G, int C, int TP, bool DROP>
__global__ __launch_bounds__(256, G <= 24 ? 2 : 1) void attention_bwd_tile_kernel(
    int n, const float *__restrict__ W1, const float *__restrict__ sc, const float *__restrict__ sh, const float *__restrict__ Ww2,
    const float *__restrict__ bw2, const float *__restrict__ v, const float *__restrict__ a, const float *__restrict__ b,
    const float *__restrict__ coord, const int *__restrict__ idx, const float *__restrict__ g_out, const float *__restrict__ Wp2,
    const float *__restrict__ bp2, float *__restrict__ gW1, float *__restrict__ part, PtvDrop drop) {
    attention_bwd_tile_body<G, C, TP, DROP>(n, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, Wp2, bp2, gW1, part, drop,
                                            (long long)blockIdx.x * TP);
}

// Whole rounds of 8-point tiles, then the remainder as 4-point tiles.  The kernel is throughput-bound per CU (two resident
// workgroups take 45 us at (24,192), one alone 24), so a last round that fills a fraction of the slots costs a whole lone
// workgroup's latency: 4 501 points = 512 + 51 tiles of 8 ran 61 -> 85 us against 4 096 points (tools/bench_bwd_tile.py --sweep).
// With the remainder cut into tiles of 4 the last round's workgroups carry half the matrix work each.
template <int G, int C, bool DROP>
__global__ __launch_bounds__(256, G <= 24 ? 2 : 1) void attention_bwd_tile_mixed_kernel(
    int n, int nfull, const float *__restrict__ W1, const float *__restrict__ sc, const float *__restrict__ sh, const float *__restrict__ Ww2,
    const float *__restrict__ bw2, const float *__restrict__ v, const float *__restrict__ a, const float *__restrict__ b,
    const float *__restrict__ coord, const int *__restrict__ idx, const float *__restrict__ g_out, const float *__restrict__ Wp2,
    const float *__restrict__ bp2, float *__restrict__ gW1, float *__restrict__ part, PtvDrop drop) {
    if ((int)blockIdx.x < nfull)
        attention_bwd_tile_body<G, C, 8, DROP>(n, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, Wp2, bp2, gW1, part, drop,
                                               (long long)blockIdx.x * 8);
    else
        attention_bwd_tile_body<G, C, 4, DROP>(n, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, Wp2, bp2, gW1, part, drop,
                                               (long long)nfull * 8 + (long long)((int)blockIdx.x - nfull) * 4);
}

template <int G, int C>
static int launch_bwd_tile_mixed(int n, int nfull, const float *W1, const float *sc, const float *sh, const float *Ww2, const float *bw2,
                                 const float *v, const float *a, const float *b, const float *coord, const int *idx, const float *g_out,
                                 const float *Wp2, const float *bp2, float *gW1, float *gsc, float *gsh, float *gWw2, float *gbw2, float *ga,
                                 float *gb, float *part, size_t part_floats_avail, PtvDrop drop, hipStream_t st) {
    using K8 = BwdTileCfg<G, C, 8>;
    using K4 = BwdTileCfg<G, C, 4>;
    static_assert(K8::PF == K4::PF, "one record format");
    const size_t lds = sizeof(float) * std::max(K8::lds_floats, K4::lds_floats);
    const bool dropping = drop.thresh != 0;
    auto kern = dropping ? attention_bwd_tile_mixed_kernel<G, C, true> : attention_bwd_tile_mixed_kernel<G, C, false>;
    static bool configured[2] = {false, false};
    if (!configured[dropping]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PTV2_ERR_LAUNCH;
        configured[dropping] = true;
    }
    const int ntail = (int)((n - (long long)nfull * 8 + 3) / 4);
    const int nblk = nfull + ntail;
    if ((size_t)nblk * K8::PF > part_floats_avail) return PTV2_ERR_WORKSPACE;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, n, nfull, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, Wp2, bp2, gW1, part, drop);
    launch_finalize(st, (const float *)part, nblk, K8::PF, MapBwdPoint{ga, gb, gsc, gsh, gWw2, gbw2, C, G});
    return PTV2_OK;
}
// whole rounds of resident workgroups the 8-point tiles fill, when what is left is at most half a round (else 0: plain launch)
template <int G, int C>
static int bwd_tile_full_rounds(int n) {
    static int slots = 0;
    if (!slots) {
        using K8 = BwdTileCfg<G, C, 8>;
        using K4 = BwdTileCfg<G, C, 4>;
        const size_t lds = sizeof(float) * std::max(K8::lds_floats, K4::lds_floats);
        int dev = 0, cus = 0, occ = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        (void)hipFuncSetAttribute((const void *)attention_bwd_tile_mixed_kernel<G, C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)attention_bwd_tile_mixed_kernel<G, C, false>, 256, lds) != hipSuccess || occ < 1) occ = 2;
        slots = occ * cus;
    }
    static const bool off = [] { const char *e = getenv("AO_AMD_BT_MIXED"); return e && e[0] == '0'; }();
    const int nt8 = (n + 7) / 8, full = nt8 / slots * slots, tail = nt8 - full;
    return (!off && full > 0 && tail > 0 && 2 * tail <= slots) ? full : 0;
}

template <int G, int C, int TP>
static int launch_bwd_tile(int n, const float *W1, const float *sc, const float *sh, const float *Ww2, const float *bw2, const float *v,
                           const float *a, const float *b, const float *coord, const int *idx, const float *g_out, const float *Wp2,
                           const float *bp2, float *gW1, float *gsc, float *gsh, float *gWw2, float *gbw2, float *ga, float *gb,
                           float *part, size_t part_floats_avail, PtvDrop drop, hipStream_t st) {
    using K = BwdTileCfg<G, C, TP>;
    const size_t lds = sizeof(float) * K::lds_floats;
    const bool dropping = drop.thresh != 0;
    auto kern = dropping ? attention_bwd_tile_kernel<G, C, TP, true> : attention_bwd_tile_kernel<G, C, TP, false>;
    static bool configured[2] = {false, false};
    if (!configured[dropping]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PTV2_ERR_LAUNCH;
        configured[dropping] = true;
    }
    const int nblk = (n + TP - 1) / TP;
    if ((size_t)nblk * K::PF > part_floats_avail) return PTV2_ERR_WORKSPACE;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, n, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, Wp2, bp2, gW1, part, drop);
    launch_finalize(st, (const float *)part, nblk, K::PF, MapBwdPoint{ga, gb, gsc, gsh, gWw2, gbw2, C, G});
    return PTV2_OK;
}

}  // namespace gva

// 1 when (k, c, g) has a tile-kernel instance
// ((64, 512), the fifth level of the ScanNet configuration -- a few dozen points -- stays on the point kernel: the instance
// needs more than 512 registers)
int gva_bwd_tile_supported(int k, int c, int g) {
    return k == 16 && ((g == 12 && c == 96) || (g == 24 && c == 192) || (g == 48 && c == 384));
}
// floats of partial records gva_bwd_tile_launch writes
size_t gva_bwd_tile_part_floats(int n, int c, int g) {
    const int tp = 4;  // (the most records any of the forms writes: tiles of 4 points)
    return (size_t)((n + tp - 1) / tp) * (4 * (size_t)c + 3 * (size_t)g + (size_t)g * g) + 64;
}

// the softmax / aggregation backward with the grouped projection's backward folded in: reads g_out (never g_A / g_sw);
// writes gW1 (n,16,g) and, through the finalize, gsc, gsh (g), gWw2 (g,g), gbw2 (g), ga (c,3), gb (c)
int gva_bwd_tile_launch(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh, const float *Ww2,
                        const float *bw2, const float *v, const float *a, const float *b, const float *coord, const int *idx,
                        const float *g_out, const float *Wp2, const float *bp2, float *gW1, float *gsc, float *gsh, float *gWw2,
                        float *gbw2, float *ga, float *gb, float *part, size_t part_floats_avail, gva::PtvDrop drop, hipStream_t st) {
    using namespace gva;
    if (!gva_bwd_tile_supported(k, c, g) || n < 1) return PTV2_ERR_ARG;
#define ARGS n, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, Wp2, bp2, gW1, gsc, gsh, gWw2, gbw2, ga, gb, part, part_floats_avail, drop, st
    if (g == 12 || g == 24) {
        const int nfull = g == 12 ? bwd_tile_full_rounds<12, 96>(n) : bwd_tile_full_rounds<24, 192>(n);
#define MARGS n, nfull, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, Wp2, bp2, gW1, gsc, gsh, gWw2, gbw2, ga, gb, part, part_floats_avail, drop, st
        if (nfull) return g == 12 ? launch_bwd_tile_mixed<12, 96>(MARGS) : launch_bwd_tile_mixed<24, 192>(MARGS);
#undef MARGS
    }
    if (g == 12) return launch_bwd_tile<12, 96, 8>(ARGS);
    if (g == 24) return launch_bwd_tile<24, 192, 8>(ARGS);
    // one workgroup per CU at this width: tiles of 4 points while they all fit in ONE round of the 256 CUs (n = 240: 61 us against
    // 80), tiles of 8 beyond (n = 1074: 135 workgroups, 86 us; 269 tiles of 4 would be two rounds, 120 us)
    if ((n + 3) / 4 <= 256) return launch_bwd_tile<48, 384, 4>(ARGS);
    return launch_bwd_tile<48, 384, 8>(ARGS);
#undef ARGS
}
