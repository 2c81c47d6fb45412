// ao_amd/csrc/gva_fwd_tile.hip -- forward of the softmax / aggregation / grouped-projection stages of grouped vector
// attention at the DEEP levels ((G, C) = (12, 96), (24, 192), (48, 384), (64, 512); K = 16) as ONE launch in which the
// (N, G, C) tensor A = w^T P never exists.
//
// Reference op: GroupedVectorAttention.forward, point_transformer_v2m2_base.py:103-129 (softmax over the neighbours, the
// einsum "n s g i, n s g -> n g i", and -- through the folded positional-encoding bias -- linear_p_bias' second Linear).
//
// Staged, this was three launches (attention_softmax_point, aggregate_tile, peb_fwd_mfma) that handed each other w (N,K,G),
// out_v (N,C) and A (N,G,C) through HBM; A alone is 1.5x (G = 24) to 3x (G = 48) the reference's own (N,K,C) tensor, and
// the three launches were 60-87 us per Block for 1-19 k points.  The grouped projection
//   out[n, 8 g + i] = sum_c' A[n, g, c'] Wp2[8 g + i, c']
// needs Wp2 (C x C: 37 KB ... 1 MB) streamed once per point unless points share it, so the unit of work is a TILE OF 16
// POINTS x a block of GB groups (a workgroup; grid = tiles x G / GB):
//   phase 1 (a wavefront per 4 points): z^T = Ww2 y^T + bw2 on the matrix cores for the block's rows, softmax over the 16 slots
//            (DPP row reductions), w / sw to memory for the backward, w^T, the relative positions and sw to LDS
//   out_v   (the same wavefront): sum_s w[s, g(o)] v[idx[s], o] for the block's 8 GB output channels, 16-byte row pieces
//   chunks  of 16 channels c', double-buffered in LDS:
//     A  a wavefront forms A[p, g, c'] = sum_s w[s, g] P[s, c'] of its 4 points (4 matrix instructions each; P = ReLU(a.pos + b)
//        is evaluated in the operand layout, its w^T / position operands stay in registers over all chunks) -> LDS
//     B  a wavefront owns GB / 4 groups: out^T (i, p) += Wp2 (8 g + i, c') A^T (c', p) over the 16 points of the tile
//        (4 matrix instructions per group and chunk; the Wp2 piece of the next chunk is in flight)
//   epilogue: out = out_v + projection + bp2 sw, and the column statistics of `out` over the tile's 16 rows (sum, sum of
//            squares about the tile mean) for the BatchNorm behind the attention.
// One workgroup barrier per chunk.  Per 16-point tile Wp2 is read once (from L2), nothing of size N G C or N K C is
// written or read.
#include <algorithm>
#include <cstdlib>

#include "gva_common.h"

namespace gva {

typedef float ft_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ft_v4f ft_mfma(float a, float b, ft_v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <int CTRL>
__device__ __forceinline__ float ft_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float ft_row16_sum(float v) {  // all-reduce over the 16 lanes that share lane >> 4
    v += ft_dpp<0xB1>(v);
    v += ft_dpp<0x4E>(v);
    v += ft_dpp<0x141>(v);
    v += ft_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float ft_row16_max(float v) {
    v = fmaxf(v, ft_dpp<0xB1>(v));
    v = fmaxf(v, ft_dpp<0x4E>(v));
    v = fmaxf(v, ft_dpp<0x141>(v));
    v = fmaxf(v, ft_dpp<0x140>(v));
    return v;
}
// (LDS hand-off inside a wavefront; a wavefront-scope fence would also wait for the global stores of w / sw: gva_bwd_tile.hip)
__device__ __forceinline__ void ft_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__host__ __device__ constexpr int ft_ww_pitch(int G) {  // >= G, = 4 mod 16
    int p = G;
    while (p % 16 != 4) ++p;
    return p;
}
__host__ __device__ constexpr int ft_point_pitch(int GB, int GP) {  // >= GB GP, = 4 mod 64: the 16 points of a ds_read_b128 on 16 slots
    int p = GB * GP;
    while (p % 64 != 4) p += 4;
    return p;
}

template <int G, int C, int GB>
struct FwdTileCfg {
    static constexpr int GTF = (G + 15) / 16, G16 = GTF * 16, GPW = ft_ww_pitch(G), QB = GB / 4, NGW = GB / 4, OB = 8 * GB,
                         NCH = C / 16, GP = 20, PP = ft_point_pitch(GB, GP), OP = OB + 4, WT = 16 * 17;
    static_assert(GB % 4 == 0 && GB <= 16 && G % GB == 0 && C == 8 * G && GPW >= G16, "group blocks of 4 q groups");
    static constexpr size_t lds_floats = 4 * (size_t)C + 4 * 256 + 16 * GPW + 16 + 2 * G16 + 16 * WT + 256 + 2 * 16 * PP + 16 * OP;
};

// stats != NULL: record [tile][2 C] = column sums of `out` over the tile's rows, sums of squares about the tile mean
// a_out != NULL: A (n, G, C) is written as well (the staged backward reads it)
template <int G, int C, int GB, bool WRITE_A, bool DROP>
__global__ __launch_bounds__(256, 2) void attention_fwd_tile_kernel(
    int n, const float *__restrict__ W1, const float *__restrict__ sc, const float *__restrict__ sh, const float *__restrict__ Ww2,
    const float *__restrict__ bw2, const float *__restrict__ v, const float *__restrict__ a, const float *__restrict__ b,
    const float *__restrict__ coord, const int *__restrict__ idx, const float *__restrict__ Wp2, const float *__restrict__ bp2,
    float *__restrict__ w, float *__restrict__ sw, float *__restrict__ out, float *__restrict__ stats, float *__restrict__ a_out,
    PtvDrop drop) {
    using K = FwdTileCfg<G, C, GB>;
    constexpr int GTF = K::GTF, G16 = K::G16, GPW = K::GPW, QB = K::QB, NGW = K::NGW, OB = K::OB, NCH = K::NCH, GP = K::GP, PP = K::PP,
                  OP = K::OP, WT = K::WT;
    extern __shared__ float4 lds4[];
    float4 *sAB = lds4;                        // [C]        (a.xyz, b)
    float4 *sPos = sAB + C;                    // [16][16]   relative positions (point, slot)
    float *sWw = (float *)(sPos + 256);        // [16][GPW]  Ww2 rows of the block, zero padded
    float *sBw = sWw + 16 * GPW;               // [16]
    float *sSc = sBw + 16;                     // [G16]
    float *sSh = sSc + G16;                    // [G16]
    float *sWt = sSh + G16;                    // [16][16 x 17]  w^T (point; row, slot)
    float *sSw = sWt + 16 * WT;                // [16][16]
    float *sA = sSw + 256;                     // [2][16][PP]    A chunk (point; group, 16 c' + pad)
    float *sOut = sA + 2 * 16 * PP;            // [16][OP]       out_v (point, channel of the block)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x, g0 = blockIdx.y * GB, o0 = 8 * g0;
    const long long last = (long long)n - 1;

    // ---- requests of phase 1 first (they travel while the parameter tables are staged): ids, logits rows
    long long pts[4];
    int srcv[4];
    float u[4][GTF][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long pt = (long long)tile * 16 + 4 * wid + i;
        pts[i] = pt < n ? pt : last;
        srcv[i] = idx[pts[i] * 16 + l15];
        const float *row = W1 + (pts[i] * 16 + l15) * G;
#pragma unroll
        for (int t = 0; t < GTF; ++t) {
            const int j0 = 16 * t + 4 * q;
            const float4 uu = *(const float4 *)(row + (j0 < G ? j0 : 0));
            u[i][t][0] = j0 < G ? uu.x : 0.f; u[i][t][1] = j0 < G ? uu.y : 0.f;
            u[i][t][2] = j0 < G ? uu.z : 0.f; u[i][t][3] = j0 < G ? uu.w : 0.f;
        }
    }
    for (int ch = tid; ch < C; ch += 256) sAB[ch] = make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]);
    for (int e = tid; e < 16 * GPW; e += 256) {
        const int row = e / GPW, j = e - row * GPW;
        sWw[e] = (row < GB && j < G) ? Ww2[(g0 + row) * G + j] : 0.f;
    }
    if (tid < 16) sBw[tid] = tid < GB ? bw2[g0 + tid] : 0.f;
    for (int j = tid; j < G16; j += 256) { sSc[j] = j < G ? sc[j] : 0.f; sSh[j] = j < G ? sh[j] : 0.f; }
    float cx[4], cy[4], cz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long ss = srcv[i] >= 0 ? srcv[i] : 0;
        cx[i] = coord[3 * ss] - coord[3 * pts[i]];
        cy[i] = coord[3 * ss + 1] - coord[3 * pts[i] + 1];
        cz[i] = coord[3 * ss + 2] - coord[3 * pts[i] + 2];
    }
    // ---- the v rows of out_v are requested now and consumed behind the softmax: lane = (slot parity lane >> 5, 16-byte piece
    //      lane & 31 of the block's 8 GB channels), eight rows per point.  The neighbour id of (point, slot) is wave-uniform
    //      (lane `slot` holds it): a scalar read, no LDS round trip in front of the gather
    const int of = lane & 31, par = lane >> 5;
    const bool fa = of < OB / 4;
    auto request_rows = [&](int i, float4 (&rows)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s0 = __builtin_amdgcn_readlane(srcv[i], 2 * j), s1 = __builtin_amdgcn_readlane(srcv[i], 2 * j + 1);
            const int src = par ? s1 : s0;
            const float *vp = (fa && src >= 0) ? v + (long long)src * C + o0 + 4 * of : ptv2_zero_pad;
            rows[j] = *(const float4 *)vp;
        }
    };
    auto reduce_rows = [&](int i, const float4 (&rows)[8]) {  // out_v of point i -> sOut
        const int p = 4 * wid + i, gl = fa ? of >> 1 : 0;
        float wg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wg[j] = sWt[p * WT + gl * 17 + 2 * j + par];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc.x = __builtin_fmaf(wg[j], rows[j].x, acc.x); acc.y = __builtin_fmaf(wg[j], rows[j].y, acc.y);
            acc.z = __builtin_fmaf(wg[j], rows[j].z, acc.z); acc.w = __builtin_fmaf(wg[j], rows[j].w, acc.w);
        }
        acc.x += __shfl_xor(acc.x, 32, WAVE); acc.y += __shfl_xor(acc.y, 32, WAVE);
        acc.z += __shfl_xor(acc.z, 32, WAVE); acc.w += __shfl_xor(acc.w, 32, WAVE);
        if (par == 0 && fa) *(float4 *)(sOut + p * OP + 4 * of) = acc;
    };
    // (two points' rows at a time: all four are 128 registers, more than the softmax leaves at two wavefronts per SIMD)
    float4 vv0[8], vv1[8];
#ifndef FT_SKIP_OUTV
    request_rows(0, vv0);
    request_rows(1, vv1);
#endif
    __syncthreads();

    // ---- phase 1: logits -> softmax over the 16 slots (= the lanes of a DPP row) for the rows of this block; the four
    //      points' matrix-instruction chains are interleaved
    ft_v4f z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = (ft_v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < GTF; ++t) {
        const float4 s4 = *(const float4 *)(sSc + 16 * t + 4 * q), h4 = *(const float4 *)(sSh + 16 * t + 4 * q);
        const float4 w4 = *(const float4 *)(sWw + l15 * GPW + 16 * t + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = ft_mfma(w4.x, fmaxf(__builtin_fmaf(s4.x, u[i][t][0], h4.x), 0.f), z[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = ft_mfma(w4.y, fmaxf(__builtin_fmaf(s4.y, u[i][t][1], h4.y), 0.f), z[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = ft_mfma(w4.z, fmaxf(__builtin_fmaf(s4.z, u[i][t][2], h4.z), 0.f), z[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = ft_mfma(w4.w, fmaxf(__builtin_fmaf(s4.w, u[i][t][3], h4.w), 0.f), z[i]);
    }
    const float4 b4 = *(const float4 *)(sBw + 4 * q);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = 4 * wid + i;
        const bool valid = srcv[i] >= 0;
        float wv[4], so[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float zz = z[i][r] + bb[r];
            const float mx = ft_row16_max(zz);
            // (the correctly rounded expf and division of the staged softmax kernel, in its order: w is bit-identical to it; the
            // ~25 vector instructions per weight do not matter at the deep levels)
            const float e = expf(zz - mx);
            const float den = ft_row16_sum(e);
            wv[r] = (valid && q < QB) ? e / den : 0.f;
            // attention dropout on the softmax output (gva_common.h: the factor is a hash of the element index, evaluated again
            // by the backward)
            if (DROP) wv[r] *= ptv2_drop_factor(drop, ((unsigned long long)pts[i] * 16 + l15) * G + (g0 + 4 * q + r));
            so[r] = ft_row16_sum(wv[r]);
            sWt[p * WT + (4 * q + r) * 17 + l15] = wv[r];
        }
        if (q < QB) {  // (rows of a clamped point repeat the last point's values)
            *(float4 *)(w + (pts[i] * 16 + l15) * G + g0 + 4 * q) = make_float4(wv[0], wv[1], wv[2], wv[3]);
            if (l15 == 0) {
                *(float4 *)(sw + pts[i] * G + g0 + 4 * q) = make_float4(so[0], so[1], so[2], so[3]);
                *(float4 *)(sSw + p * 16 + 4 * q) = make_float4(so[0], so[1], so[2], so[3]);
            }
        }
        sPos[p * 16 + l15] = valid ? make_float4(cx[i], cy[i], cz[i], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    ft_wave_sync();  // the records of my 4 points are wave-private until the first chunk barrier

    // ---- out_v of my first two points from the rows requested above; the other two points' rows travel through the chunk loop
#ifndef FT_SKIP_OUTV
    reduce_rows(0, vv0);
    reduce_rows(1, vv1);
    request_rows(2, vv0);
    request_rows(3, vv1);
#endif

    // ---- chunks of 16 channels c'
    float wA[4][4];     // A operand of phase A: w^T (row l15, slot 4 st + q) of my 4 points
    float3 pq[4][4];    // relative position of slot 4 st + q
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            wA[i][st] = sWt[(4 * wid + i) * WT + l15 * 17 + 4 * st + q];
            const float4 t = sPos[(4 * wid + i) * 16 + 4 * st + q];
            pq[i][st] = make_float3(t.x, t.y, t.z);
        }
    const float *wp2row[NGW];
    float4 wpn[NGW];
    ft_v4f acc[NGW];
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi) {
        const int g = wid + 4 * gi;
        wp2row[gi] = Wp2 + (size_t)(o0 + 8 * g + (l15 & 7)) * C + 4 * q;
        wpn[gi] = *(const float4 *)wp2row[gi];
        acc[gi] = (ft_v4f){0.f, 0.f, 0.f, 0.f};
    }
#ifdef FT_SKIP_CHUNKS
    __syncthreads();
#else
#pragma unroll 2
    for (int ck = 0; ck < NCH; ++ck) {
        float *buf = sA + (ck & 1) * 16 * PP;
        float4 wpc[NGW];
        const int cn = ck + 1 < NCH ? ck + 1 : ck;
#pragma unroll
        for (int gi = 0; gi < NGW; ++gi) { wpc[gi] = wpn[gi]; wpn[gi] = *(const float4 *)(wp2row[gi] + 16 * cn); }
        const float4 ab = sAB[16 * ck + l15];
        ft_v4f d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = (ft_v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                d[i] = ft_mfma(wA[i][st], pe_act(ab.x, ab.y, ab.z, ab.w, pq[i][st].x, pq[i][st].y, pq[i][st].z), d[i]);
        if (q < QB) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) buf[(4 * wid + i) * PP + (4 * q + r) * GP + l15] = d[i][r];
            if constexpr (WRITE_A) {  // (a template parameter: a branch in this loop costs every trip a drained memory queue)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a_out[(pts[i] * G + g0 + 4 * q + r) * C + 16 * ck + l15] = d[i][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < NGW; ++gi) {
            const float4 bx = *(const float4 *)(buf + l15 * PP + (wid + 4 * gi) * GP + 4 * q);
            acc[gi] = ft_mfma(wpc[gi].x, bx.x, acc[gi]);
            acc[gi] = ft_mfma(wpc[gi].y, bx.y, acc[gi]);
            acc[gi] = ft_mfma(wpc[gi].z, bx.z, acc[gi]);
            acc[gi] = ft_mfma(wpc[gi].w, bx.w, acc[gi]);
        }
    }
#endif

#ifndef FT_SKIP_OUTV
    reduce_rows(2, vv0);
    reduce_rows(3, vv1);
#endif
    __syncthreads();
    // ---- epilogue: D[i][p]: output 4 q + reg of the group (q < 2), point l15
    const long long ptl = (long long)tile * 16 + l15;
    const bool rv = ptl < n;
    const int cnt = (int)std::min<long long>(16, (long long)n - (long long)tile * 16);
    const float inv = 1.0f / (float)cnt;
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi) {
        const int g = wid + 4 * gi;
        if (q < 2) {
            const int ol = 8 * g + 4 * q;
            const float4 ov = *(const float4 *)(sOut + l15 * OP + ol);
            const float s = sSw[l15 * 16 + g];
            const float4 bb = *(const float4 *)(bp2 + o0 + ol);
            float4 val;
            val.x = ov.x + acc[gi][0] + bb.x * s; val.y = ov.y + acc[gi][1] + bb.y * s;
            val.z = ov.z + acc[gi][2] + bb.z * s; val.w = ov.w + acc[gi][3] + bb.w * s;
            if (rv) *(float4 *)(out + ptl * C + o0 + ol) = val;
            if (stats) {
                const float s0 = ft_row16_sum(rv ? val.x : 0.f), s1 = ft_row16_sum(rv ? val.y : 0.f);
                const float s2 = ft_row16_sum(rv ? val.z : 0.f), s3 = ft_row16_sum(rv ? val.w : 0.f);
                const float d0 = rv ? val.x - s0 * inv : 0.f, d1 = rv ? val.y - s1 * inv : 0.f;
                const float d2 = rv ? val.z - s2 * inv : 0.f, d3 = rv ? val.w - s3 * inv : 0.f;
                const float m0 = ft_row16_sum(d0 * d0), m1 = ft_row16_sum(d1 * d1), m2 = ft_row16_sum(d2 * d2), m3 = ft_row16_sum(d3 * d3);
                if (l15 == 0) {
                    float *rec = stats + (size_t)tile * 2 * C + o0 + ol;
                    *(float4 *)rec = make_float4(s0, s1, s2, s3);
                    *(float4 *)(rec + C) = make_float4(m0, m1, m2, m3);
                }
            }
        }
    }
}

template <int G, int C, int GB>
static int launch_fwd_tile(int n, const float *W1, const float *sc, const float *sh, const float *Ww2, const float *bw2, const float *v,
                           const float *a, const float *b, const float *coord, const int *idx, const float *Wp2, const float *bp2,
                           float *w, float *sw, float *out, float *stats, float *a_out, PtvDrop drop, hipStream_t st) {
    using K = FwdTileCfg<G, C, GB>;
    const size_t lds = sizeof(float) * K::lds_floats;
    const bool dropping = drop.thresh != 0;
    auto kern = a_out ? (dropping ? attention_fwd_tile_kernel<G, C, GB, true, true> : attention_fwd_tile_kernel<G, C, GB, true, false>)
                      : (dropping ? attention_fwd_tile_kernel<G, C, GB, false, true> : attention_fwd_tile_kernel<G, C, GB, false, false>);
    static bool configured[2][2] = {{false, false}, {false, false}};
    if (!configured[a_out != nullptr][dropping]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PTV2_ERR_LAUNCH;
        configured[a_out != nullptr][dropping] = true;
    }
    hipLaunchKernelGGL(kern, dim3((n + 15) / 16, G / GB), dim3(256), lds, st, n, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, Wp2, bp2, w,
                       sw, out, stats, a_out, drop);
    return PTV2_OK;
}

}  // namespace gva

// 1 when (k, c, g) has a tile-kernel instance
int gva_fwd_tile_supported(int k, int c, int g) {
    return k == 16 && ((g == 12 && c == 96) || (g == 24 && c == 192) || (g == 48 && c == 384) || (g == 64 && c == 512));
}
// rows per statistics record of gva_fwd_tile_launch
int gva_fwd_tile_stat_rows() { return 16; }

// softmax + aggregation + grouped projection of one attention forward.  stats (may be NULL): column statistics of `out` per
// 16-row tile, [ceil(n / 16)][2 c] floats (sum, sum of squares about the tile mean); a_out (may be NULL): A (n, g, c)
int gva_fwd_tile_launch(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh, const float *Ww2,
                        const float *bw2, const float *v, const float *a, const float *b, const float *coord, const int *idx,
                        const float *Wp2, const float *bp2, float *w, float *sw, float *out, float *stats, float *a_out, void *stream) {
    using namespace gva;
    if (!gva_fwd_tile_supported(k, c, g) || n < 1) return PTV2_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // W1 + idx + coord + v rows (each unique row once) in; w, sw, out out (+ Wp2 once per tile from L2: not HBM traffic)
    PtvScopedTimer t(KID_FWD_TILE + (g == 12 ? 0 : g == 24 ? 1 : g == 48 ? 2 : 3), st,
                     4.0 * ((double)n * k * (2 * g + 1) + (double)n * (3 + 2 * c + g)));
    int rc;
    const PtvDrop drop = ptv2_attn_drop_current();  // (0 outside a gva_block call with attention dropout)
#define ARGS n, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, Wp2, bp2, w, sw, out, stats, a_out, drop, st
    if (g == 12) rc = launch_fwd_tile<12, 96, 12>(ARGS);
    else if (g == 24) rc = launch_fwd_tile<24, 192, 12>(ARGS);
    else if (g == 48) rc = launch_fwd_tile<48, 384, 12>(ARGS);
    else rc = launch_fwd_tile<64, 512, 16>(ARGS);
#undef ARGS
    if (rc != PTV2_OK) return rc;
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int gva_attention_forward_hip_launcher(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh,
                                                  const float *Ww2, const float *bw2, const float *v, const float *a,
                                                  const float *b, const float *coord, const int *idx, const float *Wp2,
                                                  const float *bp2, float *w, float *sw, float *out, float *A, void *stream) {
    if (!W1 || !sc || !sh || !Ww2 || !bw2 || !v || !a || !b || !coord || !idx || !Wp2 || !bp2 || !w || !sw || !out) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    return gva_fwd_tile_launch(n, k, c, g, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, Wp2, bp2, w, sw, out, nullptr, A, stream);
}
