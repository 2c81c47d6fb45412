// ao_amd/csrc/gva_wgrad_tile.hip -- weight gradient of the grouped positional-bias projection (linear_p_bias[3], applied after
// the neighbour sum: ao_amd/ptv2/gva.py) at the deep levels, with its operand A = w^T P formed AGAIN from the saved softmax
// weights instead of read from an (N, G, C) tensor:
//
//   dWp2[8 g + i, c'] = sum_n g_out[n, 8 g + i] A[n, g, c'],   A[n, g, c'] = sum_s w[n, s, g] ReLU(a_c' . pos[n, s] + b_c')
//   dbp2[8 g + i]     = sum_n g_out[n, 8 g + i] sw[n, g]                                       (the direct part of grad bp2)
//
// Reference op: the einsum "n s g i, n s g -> n g i" of GroupedVectorAttention.forward (point_transformer_v2m2_base.py:126)
// seen from linear_p_bias[3]'s weight.  With the forward's tile kernel (gva_fwd_tile.hip) nothing of size N G C exists any
// more; the backward needs A only here, as the second operand of a product that contracts over the points, so a workgroup
// owns a block of GB groups x a range of 16 NCW channels c' -- the accumulators of that (8 GB, 16 NCW) output block live in
// registers -- and walks 16-point tiles: phase A forms the tile's A pieces exactly as the forward does (a wavefront per 4
// points, 4 matrix instructions per point and 16 channels, w^T read straight into the operand layout) and parks them in LDS;
// phase B contracts over the tile's 16 points, out^T (i, c') += g_out^T (i, p) A (p, c'), 4 matrix instructions per group and
// 16 channels.  Every workgroup leaves one partial record in the layout of the strided weight gradient ([g][8][c] weights,
// then [g][8] bias sums), so the finalize of dense.hip (per call, or batched over the deferred jobs of a backward) sums them
// in a fixed order.
#include <algorithm>
#include <cstdlib>

#include "gva_common.h"
#include "wgrad_job.h"

namespace gva {

typedef float wt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wt_v4f wt_mfma(float a, float b, wt_v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int WT_PG = 16 * 16 + 4;  // floats per (chunk, group) record of the LDS tile: 16 points x 16 channels (+ 4: quarters on distinct banks)

template <int G, int C, int GB, int NCW>
struct WgradTileCfg {
    static constexpr int NGW = GB / 4, QB = GB / 4, GS = G / GB, CR = C / (16 * NCW);
    static_assert(GB % 4 == 0 && GB <= 16 && G % GB == 0 && C == 8 * G && C % (16 * NCW) == 0, "blocks of groups x ranges of channels");
    static constexpr size_t lds_floats = (size_t)NCW * GB * WT_PG + 4 * 4 * 64;  // the A pieces + the waves' position records
};

// posrel (n, 16) float4 = masked relative positions of every (point, slot) of a job's neighbour table, written once per job
// by wp2_posrel_kernel_jobs in front of the batched launch: every (group block, channel range) workgroup that visits a tile
// reads them with one 16-byte load per lane instead of gathering neighbour coordinates behind the neighbour ids (two dependent
// round trips per tile visit).  NULL (a call outside a deferring backward): gathered here.
template <int G, int C, int GB, int NCW>
__device__ __forceinline__ void wp2_wgrad_tile_body(const dense::WgradJob &J, const int bx, const int gs, const int cr, float *lds) {
    using K = WgradTileCfg<G, C, GB, NCW>;
    constexpr int NGW = K::NGW, QB = K::QB, PG = WT_PG;
    float *sA = lds;                                                 // [NCW][GB][PG]
    float4 *sPos = (float4 *)(lds + (size_t)NCW * GB * PG);          // [4 waves][4 points][16 slots]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int n = J.n, NS = J.chunks;
    const float *__restrict__ g_out = J.gY, *__restrict__ w = J.X, *__restrict__ sw = J.rowscale;
    const float *__restrict__ coord = (const float *)J.aux[0], *__restrict__ a = (const float *)J.aux[2], *__restrict__ b = (const float *)J.aux[3];
    const int *__restrict__ idx = (const int *)J.aux[1];
    const float4 *__restrict__ posrel = (const float4 *)J.mX[0];
    const int g0 = gs * GB, o0 = 8 * g0, c0 = cr * 16 * NCW;
    const long long last = (long long)n - 1;
    float4 ab[NCW];
#pragma unroll
    for (int ck = 0; ck < NCW; ++ck) {
        const int ch = c0 + 16 * ck + l15;
        ab[ck] = make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]);
    }
    wt_v4f accW[NGW][NCW];
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
        for (int ck = 0; ck < NCW; ++ck) accW[gi][ck] = (wt_v4f){0.f, 0.f, 0.f, 0.f};
    float bacc = 0.f;
    const bool bias_thread = cr == 0 && tid < 8 * GB;
    const int bch = bias_thread ? tid : 0;
    const int lrow = l15 < GB ? l15 : GB - 1;  // (rows past the block repeat its last row: their result rows are never stored)
    const int ntiles = (n + 15) / 16;
    // operands of a tile, requested one tile ahead: w^T in the operand layout of phase A (row = group, step = slot), the
    // relative position of (my point lane >> 4, slot l15), g_out^T in the operand layout of phase B (row = output, step = point)
    struct Ops { float wA[4][4]; float4 pos; float goA[NGW][4]; };
    auto request = [&](int tile, Ops &o) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long pt = (long long)tile * 16 + 4 * wid + i, pc = pt < n ? pt : last;
#pragma unroll
            for (int st = 0; st < 4; ++st) o.wA[i][st] = w[(pc * 16 + 4 * st + q) * G + g0 + lrow];
        }
        {
            const long long pt = (long long)tile * 16 + 4 * wid + q, pc = pt < n ? pt : last;
            if (posrel) {
                o.pos = posrel[pc * 16 + l15];
            } else {
                const int sid = idx[pc * 16 + l15];
                const long long ss = sid >= 0 ? sid : 0;
                const float x = coord[3 * ss] - coord[3 * pc], y = coord[3 * ss + 1] - coord[3 * pc + 1], z = coord[3 * ss + 2] - coord[3 * pc + 2];
                o.pos = sid >= 0 ? make_float4(x, y, z, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const long long pt = (long long)tile * 16 + 4 * ks + q;
                const float val = g_out[(pt < n ? pt : last) * C + o0 + 8 * (wid + 4 * gi) + (l15 & 7)];
                o.goA[gi][ks] = pt < n ? val : 0.f;
            }
    };
    Ops nxt;
    if (bx < ntiles) request(bx, nxt);
    for (int tile = bx; tile < ntiles; tile += NS) {
        const Ops cur = nxt;
        const int tn = tile + NS < ntiles ? tile + NS : tile;
        request(tn, nxt);  // (the last trip repeats its own tile: unconditional loads)
        // ---- relative positions through a wave-private LDS record into the operand layout of phase A (slot 4 st + q)
        sPos[wid * 64 + lane] = cur.pos;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float3 pq[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float4 t = sPos[wid * 64 + i * 16 + 4 * st + q];
                pq[i][st] = make_float3(t.x, t.y, t.z);
            }
        // ---- phase A: A (group, c') of my points for the NCW chunks -> LDS [chunk][group][point][c']
#pragma unroll
        for (int ck = 0; ck < NCW; ++ck) {
            wt_v4f d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = (wt_v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    d[i] = wt_mfma(cur.wA[i][st], pe_act(ab[ck].x, ab[ck].y, ab[ck].z, ab[ck].w, pq[i][st].x, pq[i][st].y, pq[i][st].z), d[i]);
            if (q < QB) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sA[(ck * GB + 4 * q + r) * PG + (4 * wid + i) * 16 + l15] = d[i][r];
            }
        }
        if (cr == 0) {  // dbp2 of my block: thread = output channel (uniform branch; loads unconditional, masked by value)
            float gv[16], sv[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const long long pt = (long long)tile * 16 + p, pc = pt < n ? pt : last;
                gv[p] = g_out[pc * C + o0 + bch];
                sv[p] = sw[pc * G + g0 + bch / 8];
            }
#pragma unroll
            for (int p = 0; p < 16; ++p) bacc = __builtin_fmaf((long long)tile * 16 + p < n ? gv[p] : 0.f, sv[p], bacc);
        }
        __syncthreads();
        // ---- phase B: contraction over the tile's 16 points
#pragma unroll
        for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
            for (int ck = 0; ck < NCW; ++ck) {
                const float *src = sA + (ck * GB + wid + 4 * gi) * PG + q * 16 + l15;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) accW[gi][ck] = wt_mfma(cur.goA[gi][ks], src[ks * 64], accW[gi][ck]);
            }
        __syncthreads();  // the tile is rewritten by the next trip
    }
    // ---- the workgroup's record: [g][8][c] weights, [g][8] bias sums (D[i][c']: row 4 q + reg = output i of the group, q < 2)
    float *rec = J.part + (size_t)bx * J.rec;
    if (q < 2) {
#pragma unroll
        for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
            for (int ck = 0; ck < NCW; ++ck)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    rec[((size_t)(g0 + wid + 4 * gi) * 8 + 4 * q + r) * C + c0 + 16 * ck + l15] = accW[gi][ck][r];
    }
    if (bias_thread) rec[(size_t)G * 8 * C + o0 + tid] = bacc;
}

// posrel of every job in one launch (workgroup -> job through J.ldy = the job's first workgroup of THIS launch: the strided
// form's row pitch has no meaning for a recompute job)
__global__ __launch_bounds__(256) void wp2_posrel_kernel_jobs(const dense::WgradJob *__restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (long long)blockIdx.x >= jobs[j + 1].ldy) ++j;
    const dense::WgradJob &J = jobs[j];
    const long long e = ((long long)blockIdx.x - J.ldy) * 256 + threadIdx.x;
    if (e >= (long long)J.n * 16) return;
    const float *__restrict__ coord = (const float *)J.aux[0];
    const int *__restrict__ idx = (const int *)J.aux[1];
    const long long pt = e >> 4;
    const int sid = idx[e];
    const long long ss = sid >= 0 ? sid : 0;
    const float x = coord[3 * ss] - coord[3 * pt], y = coord[3 * ss + 1] - coord[3 * pt + 1], z = coord[3 * ss + 2] - coord[3 * pt + 2];
    ((float4 *)J.mX[0])[e] = sid >= 0 ? make_float4(x, y, z, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// shape dispatch (uniform over the workgroup)
__device__ __forceinline__ void wp2_wgrad_tile_dispatch(const dense::WgradJob &J, int local, float *lds) {
    const int NS = J.chunks, GS = J.tiles;
    const int bx = local % NS, rest = local / NS, gs = rest % GS, cr = rest / GS;
    switch (J.batch) {
        case 12: wp2_wgrad_tile_body<12, 96, 12, 3>(J, bx, gs, cr, lds); break;
        case 24: wp2_wgrad_tile_body<24, 192, 12, 4>(J, bx, gs, cr, lds); break;
        case 48: wp2_wgrad_tile_body<48, 384, 12, 4>(J, bx, gs, cr, lds); break;
        default: wp2_wgrad_tile_body<64, 512, 16, 4>(J, bx, gs, cr, lds); break;
    }
}

__global__ __launch_bounds__(256, 2) void wp2_wgrad_tile_kernel(dense::WgradJob J) {
    extern __shared__ float4 wt_lds4[];
    wp2_wgrad_tile_dispatch(J, (int)blockIdx.x, (float *)wt_lds4);
}

__global__ __launch_bounds__(256, 2) void wp2_wgrad_tile_kernel_jobs(const dense::WgradJob *__restrict__ jobs, int njobs) {
    extern __shared__ float4 wt_lds4[];
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].wg0) ++j;  // (uniform: scalar loads)
    const dense::WgradJob &J = jobs[j];
    wp2_wgrad_tile_dispatch(J, (int)blockIdx.x - J.wg0, (float *)wt_lds4);
}

constexpr size_t WT_LDS_BYTES = sizeof(float) * std::max({WgradTileCfg<12, 96, 12, 3>::lds_floats, WgradTileCfg<24, 192, 12, 4>::lds_floats,
                                                          WgradTileCfg<48, 384, 12, 4>::lds_floats, WgradTileCfg<64, 512, 16, 4>::lds_floats});

}  // namespace gva

// 1 when (k, c, g) has an instance
int gva_wgrad_tile_supported(int k, int c, int g) {
    return k == 16 && ((g == 12 && c == 96) || (g == 24 && c == 192) || (g == 48 && c == 384) || (g == 64 && c == 512));
}

// fills the launch geometry of a job whose operand fields are set: point splits (at most max_splits records), group blocks,
// channel ranges; returns the floats of partial records it needs
size_t gva_wgrad_tile_plan(dense::WgradJob *J, int max_splits) {
    const int g = J->batch, c = J->cin;
    const int gb = g == 64 ? 16 : 12, ncw = g == 12 ? 3 : 4;
    const int tiles = (J->n + 15) / 16;
    J->tiles = g / gb;
    J->tiles_i = c / (16 * ncw);
    J->chunks = std::max(1, std::min({tiles / 6, 48, std::max(1, max_splits)}));
    J->chunk = (tiles + J->chunks - 1) / J->chunks * 16;
    J->rec = g * (8 * c + 8);
    J->wgs = J->chunks * J->tiles * J->tiles_i;
    J->has_pb = 1;
    J->count = 0;
    J->cout = 8;
    J->gw = gb;
    return (size_t)J->chunks * J->rec;
}

static bool wgrad_tile_configure() {
    static const bool ok = hipFuncSetAttribute((const void *)gva::wp2_wgrad_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)gva::WT_LDS_BYTES) == hipSuccess &&
                           hipFuncSetAttribute((const void *)gva::wp2_wgrad_tile_kernel_jobs, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)gva::WT_LDS_BYTES) == hipSuccess;
    return ok;
}

int gva_wgrad_tile_launch_one(const dense::WgradJob &J, hipStream_t st) {
    if (!wgrad_tile_configure()) return PTV2_ERR_LAUNCH;
    hipLaunchKernelGGL(gva::wp2_wgrad_tile_kernel, dim3((unsigned)J.wgs), dim3(256), gva::WT_LDS_BYTES, st, J);
    return PTV2_OK;
}

// pos_wgs > 0: every job carries a posrel buffer (J.mX[0]) and its first workgroup of the position launch (J.ldy)
int gva_wgrad_tile_launch_jobs(const dense::WgradJob *table, int njobs, int wgs, int pos_wgs, hipStream_t st) {
    if (!wgrad_tile_configure()) return PTV2_ERR_LAUNCH;
    if (pos_wgs > 0) hipLaunchKernelGGL(gva::wp2_posrel_kernel_jobs, dim3((unsigned)pos_wgs), dim3(256), 0, st, table, njobs);
    hipLaunchKernelGGL(gva::wp2_wgrad_tile_kernel_jobs, dim3((unsigned)wgs), dim3(256), gva::WT_LDS_BYTES, st, table, njobs);
    return PTV2_OK;
}
