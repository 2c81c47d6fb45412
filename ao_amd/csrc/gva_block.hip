// ao_amd/csrc/gva_block.hip -- GroupedVectorAttention forward / backward as ONE native call each.
//
// The stage launchers (gva_fwd / gva_aggregate / gva_bwd / gva_peb / gva_fold / dense) are enqueued back to
// back on the caller's stream from here; the only additional device code is the parameter-sized glue
// (M = (Ww1 Wp2)^T, cW = Ww1 bp2 + bw1 and their gradients, a couple of vector adds, the bp2 gradient).
// With ~35 python-level ops per block the step was host-bound (profiles/r01_*: 2 350 launches, 32 ms of host
// time vs 30 ms of kernel time); behind this entry point a block costs two host calls.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "gva_common.h"
#include "gva_fold_p.h"

namespace gva {

// M[c',g] = sum_c Wp2[c,c'] Ww1[g,c];  cW[g] = sum_c Ww1[g,c] bp2[c] + bw1[g]
// one wavefront per output, lanes over the reduction index (the outputs are few, the reduction is long).
// The workgroups from index mblocks on run the (independent, equally parameter-sized) BN_p fold of the same block:
// one launch instead of two
__device__ __forceinline__ void fold_m_fwd_body(int bid, int c, int g, const float *__restrict__ Wp2,
                                                const float *__restrict__ bp2, const float *__restrict__ Ww1,
                                                const float *__restrict__ bw1, float *__restrict__ M, float *__restrict__ cW,
                                                int mblocks, const FoldPFwdArgs &P) {
    if (bid >= mblocks) {
        const int ch = (bid - mblocks) * TPB + threadIdx.x;
        if (ch < P.c) fold_p_fwd_channel(P, ch);
        return;
    }
    // A workgroup = 64 consecutive c' of one group g: lane = c' (the rows of Wp2 are read as 256 contiguous bytes), the four
    // wavefronts split the contraction index and are added through LDS in wave order.  (Round 2's form -- one wavefront per
    // output, lanes over the contraction index -- read Wp2 down a column: 64 cache lines per load, 48 us for the 15 Blocks.)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tiles_c = (c + WAVE - 1) / WAVE;
    __shared__ float s_fold[TPB / WAVE][WAVE];
    if (bid < g * tiles_c) {
        const int gi = bid / tiles_c, cp = (bid - gi * tiles_c) * WAVE + lane;
        const int cpl = cp < c ? cp : c - 1;
        const int per = (c + 3) / 4, c0 = wid * per, c1 = c0 + per < c ? c0 + per : c;
        const float *w1 = Ww1 + (size_t)gi * c;
        float acc = 0.f;
        int ci = c0;
        for (; ci + 8 <= c1; ci += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Wp2[(size_t)(ci + u) * c + cpl];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_fmaf(v[u], w1[ci + u], acc);
        }
        for (; ci < c1; ++ci) acc = __builtin_fmaf(Wp2[(size_t)ci * c + cpl], w1[ci], acc);
        s_fold[wid][lane] = acc;
        __syncthreads();
        if (wid == 0 && cp < c) M[(size_t)cp * g + gi] = ((s_fold[0][lane] + s_fold[1][lane]) + s_fold[2][lane]) + s_fold[3][lane];
    } else {
        const int gi = (bid - g * tiles_c) * (TPB / WAVE) + wid;  // one wavefront per group
        if (gi < g) {
            float acc = 0.f;
            for (int ci = lane; ci < c; ci += WAVE) acc = __builtin_fmaf(Ww1[(size_t)gi * c + ci], bp2[ci], acc);
            acc = wave_sum(acc);
            if (lane == 0) cW[gi] = acc + bw1[gi];
        }
    }
}

// workgroups of the M / cW part of fold_m_fwd_body
static int fold_m_blocks(int c, int g) { return g * ((c + WAVE - 1) / WAVE) + (g + TPB / WAVE - 1) / (TPB / WAVE); }

__global__ __launch_bounds__(TPB) void fold_m_fwd_kernel(int c, int g, const float *__restrict__ Wp2,
                                                         const float *__restrict__ bp2, const float *__restrict__ Ww1,
                                                         const float *__restrict__ bw1, float *__restrict__ M,
                                                         float *__restrict__ cW, int mblocks, FoldPFwdArgs P) {
    fold_m_fwd_body((int)blockIdx.x, c, g, Wp2, bp2, Ww1, bw1, M, cW, mblocks, P);
}

// the same for up to 8 attention blocks in one launch: these folds read parameters (and the position moments of the
// level) only, so the model runtime runs all of them ahead of the forward instead of one launch inside every Block
struct FoldFwdItem {
    int c, g, mblocks, blocks;
    const float *Wp2, *bp2, *Ww1, *bw1;
    float *M, *cW;
    FoldPFwdArgs P;
};
struct FoldFwdBatch {
    int count;
    FoldFwdItem item[8];
};
__global__ __launch_bounds__(TPB) void fold_m_fwd_batched_kernel(FoldFwdBatch B) {
    int bid = (int)blockIdx.x;
    for (int i = 0; i < B.count; ++i) {
        const FoldFwdItem &it = B.item[i];
        if (bid < it.blocks) {
            fold_m_fwd_body(bid, it.c, it.g, it.Wp2, it.bp2, it.Ww1, it.bw1, it.M, it.cW, it.mblocks, it.P);
            return;
        }
        bid -= it.blocks;
    }
}

// gWw1[g,c] (+)= sum_c' gM[c',g] Wp2[c,c'] + gcW[g] bp2[c];  gWp2[c,c'] += sum_g Ww1[g,c] gM[c',g];
// gbp2[c] += sum_g gcW[g] Ww1[g,c];  gbw1 = gcW.   gWw1 accumulates onto the two projection gradients.
struct FoldMBwdArgs {
    int c, g, mblocks, blocks;
    const float *Wp2, *bp2, *Ww1, *gM, *gcW, *gWw1_k, *gWw1_q;
    float *gWw1, *gWp2, *gbp2, *gbw1;
    FoldPBwdArgs P;
};

__device__ __forceinline__ void fold_m_bwd_body(const int bid, const FoldMBwdArgs &F) {
    const int c = F.c, g = F.g, mblocks = F.mblocks;
    const float *__restrict__ Wp2 = F.Wp2, *__restrict__ bp2 = F.bp2, *__restrict__ Ww1 = F.Ww1, *__restrict__ gM = F.gM;
    const float *__restrict__ gcW = F.gcW, *__restrict__ gWw1_k = F.gWw1_k, *__restrict__ gWw1_q = F.gWw1_q;
    float *__restrict__ gWw1 = F.gWw1, *gWp2 = F.gWp2, *gbp2 = F.gbp2, *__restrict__ gbw1 = F.gbw1;
    const FoldPBwdArgs &P = F.P;
    if (bid >= mblocks) {  // the BN_p fold backward of the same block rides along (see fold_m_fwd_kernel)
        const int ch = (bid - mblocks) * TPB + threadIdx.x;
        if (ch < P.c) fold_p_bwd_channel(P, ch);
        return;
    }
    // Both products walk gM (c',g) along g and a weight matrix along its rows, so that a wavefront's loads are whole cache
    // lines: with the lanes along c' (stride g floats) every load instruction touched 64 lines -- 7 M line requests for
    // the 14 MFLOP of a C = 384 block, 44 us.
    const long long n1 = (long long)g * c, n2 = (long long)c * c;
    const bool wide = c >= 256;  // narrow levels: one wavefront per gWw1 output (short chains, the strided column is small)
    const int wblocks = wide ? (int)((n1 + TPB - 1) / TPB) : (int)((n1 + WPB - 1) / WPB);
    if (bid < wblocks && !wide) {
        const long long o = (long long)bid * WPB + (threadIdx.x >> 6);
        if (o < n1) {
            const int lane = threadIdx.x & 63;
            const int gi = (int)(o / c), ci = (int)(o - (long long)gi * c);
            float acc = 0.f;
            for (int cp = lane; cp < c; cp += WAVE) acc = __builtin_fmaf(gM[(size_t)cp * g + gi], Wp2[(size_t)ci * c + cp], acc);
            acc = wave_sum(acc);
            if (lane == 0) gWw1[o] = acc + gcW[gi] * bp2[ci] + gWw1_k[o] + gWw1_q[o];
        }
        return;
    }
    if (bid < wblocks) {  // wide: one thread per gWw1 output, g fastest
        // 16 independent chains per thread, all 32 loads of a trip in flight: with 4 chains the loop was c / 4 dependent L2
        // round trips (28 us at C = 384 for 14 MFLOP, profiles/r02_final_step_sequence.txt)
        const long long o = (long long)bid * TPB + threadIdx.x;
        if (o < n1) {
            const int ci = (int)(o / g), gi = (int)(o - (long long)ci * g);
            const float *wrow = Wp2 + (size_t)ci * c;  // shared by the g threads of this ci (broadcast)
            float acc[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u] = 0.f;
            int cp = 0;
            for (; cp + 15 < c; cp += 16) {
                float m[16], w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { m[u] = gM[(size_t)(cp + u) * g + gi]; w[u] = wrow[cp + u]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u] = __builtin_fmaf(m[u], w[u], acc[u]);
            }
            for (; cp < c; ++cp) acc[0] = __builtin_fmaf(gM[(size_t)cp * g + gi], wrow[cp], acc[0]);
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) t += acc[u];
            const size_t oo = (size_t)gi * c + ci;
            gWw1[oo] = t + gcW[gi] * bp2[ci] + gWw1_k[oo] + gWw1_q[oo];
        }
        return;
    }
    const long long e = n1 + (long long)(bid - wblocks) * TPB + threadIdx.x;
    if (false) {
    } else if (e < n1 + n2) {
        const long long r = e - n1;
        const int cp = (int)(r / c), ci = (int)(r - (long long)cp * c);  // ci fastest: Ww1 rows coalesced, gM row broadcast
        const float *mrow = gM + (size_t)cp * g;
        const float prev = gWp2[(size_t)ci * c + cp];  // (requested before the loop, not behind it)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // g is a multiple of 2; four chains keep the loads of a trip in flight
        int gi = 0;
        for (; gi + 3 < g; gi += 4) {
            a0 = __builtin_fmaf(Ww1[(size_t)gi * c + ci], mrow[gi], a0);
            a1 = __builtin_fmaf(Ww1[(size_t)(gi + 1) * c + ci], mrow[gi + 1], a1);
            a2 = __builtin_fmaf(Ww1[(size_t)(gi + 2) * c + ci], mrow[gi + 2], a2);
            a3 = __builtin_fmaf(Ww1[(size_t)(gi + 3) * c + ci], mrow[gi + 3], a3);
        }
        for (; gi < g; ++gi) a0 = __builtin_fmaf(Ww1[(size_t)gi * c + ci], mrow[gi], a0);
        gWp2[(size_t)ci * c + cp] = prev + ((a0 + a1) + (a2 + a3));
    } else if (e < n1 + n2 + c) {
        const int ci = (int)(e - n1 - n2);
        float acc = gbp2[ci];
        for (int gi = 0; gi < g; ++gi) acc = __builtin_fmaf(gcW[gi], Ww1[(size_t)gi * c + ci], acc);
        gbp2[ci] = acc;
    } else if (e < n1 + n2 + c + g) {
        const int gi = (int)(e - n1 - n2 - c);
        gbw1[gi] = gcW[gi];
    }
}

__global__ __launch_bounds__(TPB) void fold_m_bwd_kernel(FoldMBwdArgs F) { fold_m_bwd_body((int)blockIdx.x, F); }

// the same for up to 8 attention blocks in one launch: these glue sums produce parameter gradients only (nothing on the
// backward's chain reads them), so the model runtime queues them and runs all of them at the end of the backward
struct FoldMBwdBatch {
    int count;
    FoldMBwdArgs item[8];
};
__global__ __launch_bounds__(TPB) void fold_m_bwd_batched_kernel(FoldMBwdBatch B) {
    int bid = (int)blockIdx.x;
    for (int i = 0; i < B.count; ++i) {
        if (bid < B.item[i].blocks) {
            fold_m_bwd_body(bid, B.item[i]);
            return;
        }
        bid -= B.item[i].blocks;
    }
}

// partial[blk][c] = sum over the block's rows of g_out[n,c] * sw[n, c / I]
__global__ __launch_bounds__(TPB) void bp2_grad_kernel(int n, int c, int g, const float *__restrict__ g_out,
                                                       const float *__restrict__ sw, float *part, unsigned *counter,
                                                       float *__restrict__ gbp2) {
    extern __shared__ float lds[];
    const int I = c / g;
    const int rl = TPB / c > 0 ? TPB / c : 1;
    const int ch0 = threadIdx.x % c, r = threadIdx.x / c;
    for (int cb = 0; cb < c; cb += TPB) {  // channels beyond 256 in further passes
        const int ch = cb + (c >= TPB ? threadIdx.x : ch0);
        float acc = 0.f;
        if (ch < c && (c >= TPB || r < rl)) {
            // (a lane has ~8 rows: all of their loads in flight at once; row by row the kernel was parked on memory for 81 %
            // of its wave cycles)
            const long long step = (long long)gridDim.x * rl;
            long long row = (long long)blockIdx.x * rl + (c >= TPB ? 0 : r);
            for (; row + 3 * step < n; row += 4 * step) {
                const float g0 = g_out[row * c + ch], g1 = g_out[(row + step) * c + ch];
                const float g2 = g_out[(row + 2 * step) * c + ch], g3 = g_out[(row + 3 * step) * c + ch];
                const float s0 = sw[row * g + ch / I], s1 = sw[(row + step) * g + ch / I];
                const float s2 = sw[(row + 2 * step) * g + ch / I], s3 = sw[(row + 3 * step) * g + ch / I];
                acc = __builtin_fmaf(g0, s0, acc); acc = __builtin_fmaf(g1, s1, acc);
                acc = __builtin_fmaf(g2, s2, acc); acc = __builtin_fmaf(g3, s3, acc);
            }
            for (; row < n; row += step) acc = __builtin_fmaf(g_out[row * c + ch], sw[row * g + ch / I], acc);
        }
        lds[threadIdx.x] = acc;
        __syncthreads();
        if (c >= TPB) {
            if (ch < c) part_store(part + (size_t)blockIdx.x * c + ch, acc);
        } else if (threadIdx.x < c) {
            float t = 0.f;
            for (int kk = 0; kk < rl; ++kk) t += lds[kk * c + threadIdx.x];
            part_store(part + (size_t)blockIdx.x * c + threadIdx.x, t);
        }
        __syncthreads();
        if (c < TPB) break;
    }
    if (counter && last_block_arrives(counter)) finalize_columns(part, gridDim.x, c, MapVec<float>{gbp2});
}

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace gva

using namespace gva;

extern "C" {
size_t gva_workspace_bytes(int n, int k, int c, int g);
size_t gva_aggregate_workspace_bytes(int n, int k, int c, int g);
size_t dense_workspace_bytes(int n, int cout, int cin);
int gva_fold_p_forward_hip_launcher(int, const float *, const float *, const float *, const float *, const double *,
                                    const double *, float *, float *, long long *, int, double, float, float, float *,
                                    float *, float *, void *);
int gva_fold_p_backward_hip_launcher(int, const float *, const float *, const float *, const double *, const double *,
                                     const float *, const float *, int, const float *, const float *, float *, float *,
                                     float *, float *, void *);
int gva_fold_w_forward_hip_launcher(int, const double *, const double *, const float *, const float *, float *, float *,
                                    long long *, int, double, float, float, float *, float *, double *, double *, void *);
int gva_fold_w_backward_hip_launcher(int, const float *, const double *, const double *, int, double, const float *,
                                     const float *, double *, double *, float *, float *, void *);
int skinny_linear_forward_hip_launcher(int, int, int, const float *, const float *, float *, void *);
int skinny_linear_backward_hip_launcher(int, int, int, const float *, const float *, float *, void *);
int linear_wgrad_hip_launcher(int, int, int, const float *, const float *, float *, float *, void *, size_t, void *);
int linear_wgrad_strided_rowscale(int, int, int, int, const float *, long long, long long, const float *, long long, long long,
                                  float *, float *, const float *, long long, int *, void *, size_t, void *);
int linear_wgrad_strided_hip_launcher(int, int, int, int, const float *, long long, long long, const float *, long long,
                                      long long, float *, float *, void *, size_t, void *);
int gva_logits_forward_hip_launcher(int, int, int, int, const float *, const float *, const float *, const float *,
                                    const float *, const float *, const float *, const int *, float *, double *, double *,
                                    void *, size_t, void *);
int gva_logits_backward_hip_launcher(int, int, int, int, const float *, const float *, const float *, const float *,
                                     const int *, const float *, const float *, const double *, const double *,
                                     const int *, const int *, float *, float *, float *, float *, float *, float *, void *,
                                     size_t, void *);
int gva_aggregate_forward_hip_launcher(int, int, int, int, const float *, const float *, const float *, const float *,
                                       const float *, const float *, const float *, const float *, const float *,
                                       const int *, float *, float *, float *, float *, void *);
int gva_aggregate_backward_hip_launcher(int, int, int, int, const float *, const float *, const float *, const float *,
                                        const float *, const float *, const float *, const float *, const float *,
                                        const int *, const float *, const float *, const float *, const float *,
                                        const int *, const int *, float *, float *, float *, float *, float *, float *,
                                        float *, float *, void *, size_t, void *);
int gva_peb_forward_hip_launcher(int, int, int, const float *, const float *, const float *, const float *, const float *,
                                 float *, void *);
int gva_peb_backward_hip_launcher(int, int, int, const float *, const float *, const float *, float *, float *, void *);
}

int gva_logits_forward_fold(int n, int k, int c, int g, const float *kW, const float *qW, const float *a, const float *b,
                            const float *M, const float *cW, const float *coord, const int *idx, float *W1, double *T1, double *T2,
                            const gva::FoldWFwdArgs &F, void *workspace, size_t workspace_bytes, void *stream);
int skinny_linear_forward_pair(int n, int cin, int cout, const float *const *x, const float *W, const float *const *xsc,
                               const float *const *xsh, float *const *y, void *stream);
int skinny_backward_pair_bn_reduce(int n, int cin, int cout, const float *const *gy, const float *W, float *const *gx, void *stream);
int gva_bwd_point_local(int k, int c, int g);
int gva_bwd_tile_path(int k, int c, int g);
int gva_fwd_point_supported(int k, int c, int g);
int gva_fwd_point_max_n();
int gva_fwd_point_launch(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh, const float *Ww2,
                         const float *bw2, const float *v, const float *a, const float *b, const float *coord, const int *idx,
                         const float *Wp2, const float *bp2, float *w, float *sw, float *A, float *out, float *stats,
                         void *stream);
int gva_fwd_tile_supported(int k, int c, int g);
int gva_wp2_wgrad_recompute(int n, int k, int c, int g, const float *g_out, const float *w, const float *sw, const float *a,
                            const float *b, const float *coord, const int *idx, float *dW, float *db, void *workspace,
                            size_t workspace_bytes, void *stream);
// the deep levels' one-launch forward (gva_fwd_tile.hip) is the path of this shape; AO_AMD_FWD_STAGED: the three staged launches
static bool gva_tile_path(int k, int c, int g) { return gva_fwd_tile_supported(k, c, g) && !getenv("AO_AMD_FWD_STAGED"); }
// 1 when the forward of this shape writes A (n, g, c) for the backward (block.hip sizes the saved buffer with it): the
// staged launches, the full-resolution point kernel, or the tile path with AO_AMD_TILE_KEEP_A (the A-reading weight gradient)
int gva_block_keeps_A(int k, int c, int g) { return !gva_tile_path(k, c, g) || getenv("AO_AMD_TILE_KEEP_A") != nullptr; }
int gva_fwd_tile_launch(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh, const float *Ww2,
                        const float *bw2, const float *v, const float *a, const float *b, const float *coord, const int *idx,
                        const float *Wp2, const float *bp2, float *w, float *sw, float *out, float *stats, float *a_out, void *stream);
int gva_logits_backward_foldw(int n, int k, int c, int g, const float *a, const float *b, const float *M, const float *coord,
                              const int *idx, const float *W1, const float *gW1, const double *gT1, const double *gT2,
                              const gva::FoldWBwdArgs &F, const int *inv_ptr, const int *inv_rows, float *gkW, float *gqW, float *ga,
                              float *gb, float *gM, float *gcW, void *workspace, size_t workspace_bytes, void *stream);
int gva_aggregate_backward_fused_peb(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh,
                                     const float *Ww2, const float *bw2, const float *v, const float *a, const float *b,
                                     const float *coord, const int *idx, const float *w, const float *g_out, const float *Wp2,
                                     const float *bp2, const int *inv_ptr, const int *inv_rows, float *gW1, float *gsc,
                                     float *gsh, float *gWw2, float *gbw2, float *gv, float *ga, float *gb, void *workspace,
                                     size_t workspace_bytes, void *stream);

namespace {
struct BlockWs {  // carve the block workspace
    char *stage; size_t stage_bytes;       // scratch handed to the stage launchers
    float *out_v, *gA, *g_sw, *gW1, *gkW, *gqW, *ga2, *gb2, *ga1, *gb1, *gM, *gcW, *gsc, *gsh, *gWw1_k, *gWw1_q, *part;
    char *wp2_part, *kq_part; size_t wp2_bytes, kq_bytes;  // split-K records whose sums ride on a later launch (riders)
    double *T1, *T2, *gT1, *gT2;
    size_t bytes;
};

BlockWs carve(void *base, int n, int k, int c, int g) {
    BlockWs w;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += al(bytes); return r; };
    const size_t rows = (size_t)n * k;
    w.stage_bytes = std::max({gva_workspace_bytes(n, k, c, g), gva_aggregate_workspace_bytes(n, k, c, g),
                              dense_workspace_bytes(n, c, c), dense_workspace_bytes(n, 2 * g, c),
                              dense_workspace_bytes((int)std::min<size_t>(rows, 2147483647), g, g)});
    w.stage = take(w.stage_bytes);
    w.out_v = (float *)take(sizeof(float) * (size_t)n * c);
    // (g_A (n,g,c) / g_sw only where a peb_bwd launch hands them to the aggregation backward: not at the full-resolution level's
    // point kernel nor on the deep levels' tile path, which form them on chip)
    const bool fused_peb = (gva_bwd_point_local(k, c, g) && !getenv("AO_AMD_BWD_STAGED")) || gva_bwd_tile_path(k, c, g);
    w.gA = fused_peb ? nullptr : (float *)take(sizeof(float) * (size_t)n * g * c);
    w.g_sw = fused_peb ? nullptr : (float *)take(sizeof(float) * (size_t)n * g);
    w.gW1 = (float *)take(sizeof(float) * rows * g);
    w.gkW = (float *)take(sizeof(float) * (size_t)n * g);
    w.gqW = (float *)take(sizeof(float) * (size_t)n * g);
    w.ga1 = (float *)take(sizeof(float) * 3 * c);
    w.gb1 = (float *)take(sizeof(float) * c);
    w.ga2 = (float *)take(sizeof(float) * 3 * c);
    w.gb2 = (float *)take(sizeof(float) * c);
    w.gM = (float *)take(sizeof(float) * (size_t)c * g);
    w.gcW = (float *)take(sizeof(float) * g);
    w.gsc = (float *)take(sizeof(float) * g);
    w.gsh = (float *)take(sizeof(float) * g);
    w.gWw1_k = (float *)take(sizeof(float) * (size_t)g * c);
    w.gWw1_q = (float *)take(sizeof(float) * (size_t)g * c);
    w.part = (float *)take(sizeof(float) * (size_t)MAX_BLOCKS * c);
    w.wp2_bytes = dense_workspace_bytes(n, c, c);
    w.wp2_part = take(w.wp2_bytes);
    w.kq_bytes = dense_workspace_bytes(n, 2 * g, c);
    w.kq_part = take(w.kq_bytes);
    w.T1 = (double *)take(sizeof(double) * g);
    w.T2 = (double *)take(sizeof(double) * g);
    w.gT1 = (double *)take(sizeof(double) * g);
    w.gT2 = (double *)take(sizeof(double) * g);
    w.bytes = off;
    return w;
}
}  // namespace

#define RUN(call)                  \
    do {                           \
        int rc_ = (call);          \
        if (rc_ != PTV2_OK) return rc_; \
    } while (0)

extern "C" size_t gva_block_workspace_bytes(int n, int k, int c, int g) {
    if (n < 0 || k < 1 || c < 1 || g < 1) return 0;
    return carve(nullptr, n, k, c, g).bytes + 1024;
}

namespace {
thread_local int g_prefolded = 0;
// deferred M / cW glue (model.hip): while a scratch region is set, gva_block_backward keeps its parameter-sized sums there
// (instead of in the shared workspace) and queues the glue launch; ptv2_gva_flush_folds() runs the queue in batches of 8
thread_local float *g_fold_scratch = nullptr;
thread_local std::vector<gva::FoldMBwdArgs> *g_fold_queue = nullptr;
}  // namespace
size_t ptv2_gva_fold_scratch_floats(int c, int g) {
    return gva::al(sizeof(float) * ((size_t)c * g + g + 2 * (size_t)g * c + 8 * (size_t)c + 64)) / sizeof(float);
}
void ptv2_gva_set_fold_scratch(float *p) { g_fold_scratch = p; }
int ptv2_gva_flush_folds(void *stream) {
    if (!g_fold_queue || g_fold_queue->empty()) return PTV2_OK;
    std::vector<gva::FoldMBwdArgs> &Q = *g_fold_queue;
    for (size_t i0 = 0; i0 < Q.size(); i0 += 8) {
        gva::FoldMBwdBatch batch{};
        int total = 0;
        for (size_t i = i0; i < Q.size() && i < i0 + 8; ++i) {
            batch.item[batch.count++] = Q[i];
            total += Q[i].blocks;
        }
        hipLaunchKernelGGL(gva::fold_m_bwd_batched_kernel, dim3(total), dim3(gva::TPB), 0, (hipStream_t)stream, batch);
    }
    Q.clear();
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
void ptv2_gva_drop_folds() {
    if (g_fold_queue) g_fold_queue->clear();
}
// internal to the library (block.hip / model.hip): the folds of the blocks that follow have been run by
// gva_fold_forward_batched for this forward already
void ptv2_gva_set_prefolded(int on) { g_prefolded = on; }

static FoldPFwdArgs fold_p_args(const ptv2_gva_block *B) {
    return FoldPFwdArgs{B->c, B->Wp1, B->bp1, B->gamma_p, B->beta_p, B->mu, B->cov, B->run_mean_p, B->run_var_p, B->batches_p,
                        B->training, (double)B->n * B->k, B->eps_p, B->momentum_p, B->a, B->b, B->rstd_p};
}

int gva_fold_forward_batched(int count, const ptv2_gva_block *blocks, void *stream) {
    for (int i0 = 0; i0 < count; i0 += 8) {
        FoldFwdBatch batch{};
        int total = 0;
        for (int i = i0; i < count && i < i0 + 8; ++i) {
            const ptv2_gva_block *B = blocks + i;
            if (B->n < 1 || B->c < 4 || B->g < 1) return PTV2_ERR_ARG;
            FoldFwdItem &it = batch.item[batch.count++];
            it.c = B->c; it.g = B->g;
            it.mblocks = fold_m_blocks(B->c, B->g);
            it.blocks = it.mblocks + divup(B->c, TPB);
            it.Wp2 = B->Wp2; it.bp2 = B->bp2; it.Ww1 = B->Ww1; it.bw1 = B->bw1; it.M = B->M; it.cW = B->cW;
            it.P = fold_p_args(B);
            total += it.blocks;
        }
        hipLaunchKernelGGL(fold_m_fwd_batched_kernel, dim3(total), dim3(TPB), 0, (hipStream_t)stream, batch);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

int gva_peb_forward_stats(int n, int c, int g, const float *A, const float *Wp2, const float *bp2, const float *sw,
                          const float *out_v, float *out, float *stats, int *stats_done, void *stream);
int gva_block_forward_stats(const ptv2_gva_block *B, float *out_stats, int *stats_done, void *workspace, size_t workspace_bytes,
                            void *stream);

extern "C" int gva_block_forward_hip_launcher(const ptv2_gva_block *B, void *workspace, size_t workspace_bytes,
                                              void *stream) {
    return gva_block_forward_stats(B, nullptr, nullptr, workspace, workspace_bytes, stream);
}

// internal (block.hip): out_stats != NULL asks the last stage for the column statistics of `out` per block of rows (the
// BatchNorm behind the attention then needs no statistics pass); *stats_done = rows per record (64, or 16 from the tile
// kernel: the buffer holds bn_tiles_floats_rb(n, c, 16) floats), 0 when none were produced
int gva_block_forward_stats(const ptv2_gva_block *B, float *out_stats, int *stats_done, void *workspace, size_t workspace_bytes,
                            void *stream) {
    if (stats_done) *stats_done = 0;
    if (!B) return PTV2_ERR_ARG;
    const int n = B->n, k = B->k, c = B->c, g = B->g;
    if (n < 0 || k < 1 || c < 4 || g < 1 || c % g != 0) return PTV2_ERR_ARG;
    if (B->attn_drop_p < 0.f || B->attn_drop_p > 1.f) return PTV2_ERR_ARG;
    const gva::PtvAttnDropScope attn_drop(B->training ? B->attn_drop_p : 0.f, B->attn_drop_seed);
    if (n == 0) return PTV2_OK;
    BlockWs W = carve(workspace, n, k, c, g);
    if (!workspace || workspace_bytes < W.bytes) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const double rows = (double)n * k;
    if (!g_prefolded) {
        const int mblocks = fold_m_blocks(c, g);
        hipLaunchKernelGGL(fold_m_fwd_kernel, dim3(mblocks + divup(c, TPB)), dim3(TPB), 0, st, c, g, B->Wp2, B->bp2, B->Ww1, B->bw1,
                           B->M, B->cW, mblocks, fold_p_args(B));
    }
    {
        const float *xs[2] = {B->key, B->q}, *xsc[2] = {B->k_sc, B->q_sc}, *xsh[2] = {B->k_sh, B->q_sh};
        float *ys[2] = {B->kW, B->qW};
        RUN(skinny_linear_forward_pair(n, c, g, xs, B->Ww1, xsc, xsh, ys, stream));
    }
    // logits + their BatchNorm statistics; the final reduction of the sums also folds BN_w into (sc, sh)
    RUN(gva_logits_forward_fold(n, k, c, g, B->kW, B->qW, B->a, B->b, B->M, B->cW, B->coord, B->idx, B->W1, W.T1, W.T2,
                                FoldWFwdArgs{B->gamma_w, B->beta_w, B->run_mean_w, B->run_var_w, B->batches_w, B->training, rows,
                                             B->eps_w, B->momentum_w, B->sc, B->sh, B->mean_w, B->rstd_w},
                                W.stage, W.stage_bytes, stream));
    // softmax, aggregation and the grouped projection: one launch at the full-resolution level (gva_fwd_point.hip), else three
    if (gva_fwd_point_supported(k, c, g) && n <= gva_fwd_point_max_n() && !gva::ptv2_attn_drop_current().thresh &&
        !getenv("AO_AMD_FWD_STAGED")) {
        RUN(gva_fwd_point_launch(n, k, c, g, B->W1, B->sc, B->sh, B->Ww2, B->bw2, B->v, B->a, B->b, B->coord, B->idx, B->Wp2, B->bp2,
                                 B->w, B->sw, B->A, B->out, out_stats, stream));
        if (out_stats && stats_done) *stats_done = 64;
    } else if (gva_tile_path(k, c, g)) {
        // the deep levels: one launch per 16-point tile x group block (gva_fwd_tile.hip); no out_v, no A
        RUN(gva_fwd_tile_launch(n, k, c, g, B->W1, B->sc, B->sh, B->Ww2, B->bw2, B->v, B->a, B->b, B->coord, B->idx, B->Wp2, B->bp2,
                                B->w, B->sw, B->out, out_stats, gva_block_keeps_A(k, c, g) ? B->A : nullptr, stream));
        if (out_stats && stats_done) *stats_done = 16;
    } else {
        RUN(gva_aggregate_forward_hip_launcher(n, k, c, g, B->W1, B->sc, B->sh, B->Ww2, B->bw2, B->v, B->a, B->b, B->coord,
                                               B->idx, W.out_v, B->A, B->sw, B->w, stream));
        RUN(gva_peb_forward_stats(n, c, g, B->A, B->Wp2, B->bp2, B->sw, W.out_v, B->out, out_stats, stats_done, stream));
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int gva_block_backward_hip_launcher(const ptv2_gva_block *B, const ptv2_gva_block_grads *G, void *workspace,
                                               size_t workspace_bytes, void *stream) {
    if (!B || !G) return PTV2_ERR_ARG;
    const int n = B->n, k = B->k, c = B->c, g = B->g;
    if (n < 0 || k < 1 || c < 4 || g < 1 || c % g != 0) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const gva::PtvAttnDropScope attn_drop(B->training ? B->attn_drop_p : 0.f, B->attn_drop_seed);
    BlockWs W = carve(workspace, n, k, c, g);
    if (!workspace || workspace_bytes < W.bytes) return PTV2_ERR_WORKSPACE;
    if (g_fold_scratch) {  // the glue's operands in the caller's per-Block region: they outlive this call
        float *p = g_fold_scratch;
        W.gM = p; p += (size_t)c * g;
        W.gcW = p; p += g;
        W.gWw1_k = p; p += (size_t)g * c;
        W.gWw1_q = p; p += (size_t)g * c;
        W.ga1 = p; p += 3 * (size_t)c;
        W.gb1 = p; p += c;
        W.ga2 = p; p += 3 * (size_t)c;
        W.gb2 = p;
    }
    // inside a model backward that defers weight gradients (dense.hip: WgradJob; the caller kept g_out): gkW / gqW, the operands
    // of the kW / qW weight gradient, go to the deferral arena too (its results already live in the per-Block glue region)
    bool kq_kept = false;
    if (g_fold_scratch && ptv2_wgrad_defer_armed_rs()) {
        float *a = ptv2_wgrad_defer_alloc((size_t)n * g), *b2 = a ? ptv2_wgrad_defer_alloc((size_t)n * g) : nullptr;
        if (a && b2) { W.gkW = a; W.gqW = b2; kq_kept = true; }
    }
    hipStream_t st = (hipStream_t)stream;
    const double rows = (double)n * k;
    const int I = c / g;
    PtvRiderGuard riders;  // an error return below must not leave queued sums (pointers into this call's workspace) behind
    // 1. projection after the neighbour sum: g_A, g_sw (formed inside the point kernel for the narrow instances),
    //    grad Wp2 (direct part), grad bp2 (direct part)
    const bool fused_peb = (gva_bwd_point_local(k, c, g) && !getenv("AO_AMD_BWD_STAGED")) || gva_bwd_tile_path(k, c, g);
    if (fused_peb && !G->inv_ptr) return PTV2_ERR_ARG;  // (the fused forms gather grad v through the inverse neighbour table)
    if (!fused_peb) RUN(gva_peb_backward_hip_launcher(n, c, g, G->g_out, B->Wp2, B->bp2, W.gA, W.g_sw, stream));
    int bp2_done = 0;
    {
        // the sum of its split-K records (needed by fold_m_bwd only) rides on the gv launch of the aggregation stage: records
        // in a region of their own, which nothing before that launch overwrites.  The same launch forms the direct part of
        // grad bp2 = sum_n g_out[n, ch] sw[n, group(ch)] as its weighted bias sums (it reads g_out anyway; that sum was a launch
        // of its own per Block, bp2_grad_kernel: 15 x 6 us); where the weight gradient cannot (bf16 operands), the kernel below
        const PtvDeferScope defer;
        if (!gva_block_keeps_A(k, c, g)) {  // A = w^T P is formed again inside the weight gradient (gva_wgrad_tile.hip)
            RUN(gva_wp2_wgrad_recompute(n, k, c, g, G->g_out, B->w, B->sw, B->a, B->b, B->coord, B->idx, G->gWp2, G->gbp2, W.wp2_part,
                                        W.wp2_bytes, stream));
            bp2_done = 1;
        } else
        RUN(linear_wgrad_strided_rowscale(n, I, c, g, G->g_out, c, I, B->A, (long long)g * c, c, G->gWp2, G->gbp2, B->sw, g, &bp2_done,
                                          W.wp2_part, W.wp2_bytes, stream));
    }
    if (!bp2_done) {
        const PtvDeferScope defer;  // (its record sum rides on the gv launch as well)
        const int rl = std::max(1, TPB / c);
        const int nblk = (int)std::min<long long>(((long long)n + rl * 8 - 1) / (rl * 8), MAX_BLOCKS);
        const bool own_final = (size_t)nblk * c <= FUSED_FINAL_MAX;
        unsigned *cnt = own_final ? ptv2_stream_counters(st) : nullptr;
        if (own_final && !cnt) return PTV2_ERR_LAUNCH;
        hipLaunchKernelGGL(bp2_grad_kernel, dim3(nblk), dim3(TPB), sizeof(float) * TPB, st, n, c, g, G->g_out, B->sw, W.part,
                           cnt ? cnt + CNT_BP2 : nullptr, G->gbp2);
        if (!own_final) launch_finalize(st, (const float *)W.part, nblk, c, MapVec<float>{G->gbp2});
    }
    // 2. softmax / aggregation stage
    if (fused_peb)
        RUN(gva_aggregate_backward_fused_peb(n, k, c, g, B->W1, B->sc, B->sh, B->Ww2, B->bw2, B->v, B->a, B->b, B->coord, B->idx,
                                             B->w, G->g_out, B->Wp2, B->bp2, G->inv_ptr, G->inv_rows, W.gW1, W.gsc, W.gsh,
                                             G->gWw2, G->gbw2, G->gv, W.ga2, W.gb2, W.stage, W.stage_bytes, stream));
    else
        RUN(gva_aggregate_backward_hip_launcher(n, k, c, g, B->W1, B->sc, B->sh, B->Ww2, B->bw2, B->v, B->a, B->b, B->coord,
                                                B->idx, B->w, G->g_out, W.gA, W.g_sw, G->inv_ptr, G->inv_rows, W.gW1, W.gsc,
                                                W.gsh, G->gWw2, G->gbw2, G->gv, W.ga2, W.gb2, W.stage, W.stage_bytes, stream));
    // 3. + 4. BatchNorm over the logits (its backward is evaluated in the prologue of the rows kernel), logits stage
    if (!G->inv_ptr) (void)ptv2_zero_async(W.gkW, sizeof(float) * (size_t)n * g, st);
    // the parameter-gradient sums of this stage and of the kW / qW weight gradient ride on the skinny_bwd launch below
    {
    const PtvDeferScope defer;
    RUN(gva_logits_backward_foldw(n, k, c, g, B->a, B->b, B->M, B->coord, B->idx, B->W1, W.gW1, nullptr, nullptr,
                                  FoldWBwdArgs{B->gamma_w, B->mean_w, B->rstd_w, B->training, rows, W.gsc, W.gsh, G->ggamma_w,
                                               G->gbeta_w},
                                  G->inv_ptr, G->inv_rows, W.gkW, W.gqW, W.ga1, W.gb1, W.gM, W.gcW, W.stage, W.stage_bytes, stream));
    // 6. projections kW = k Ww1^T, qW = q Ww1^T: weight gradient first (its records in a region of their own), then the
    //    input gradients -- the launch that carries the queued sums
    {
        const float *gys[2] = {W.gkW, W.gqW}, *xs[2] = {B->key, B->q};
        float *dws[2] = {W.gWw1_k, W.gWw1_q};
        const float *xsc[2] = {B->k_sc, B->q_sc}, *xsh[2] = {B->k_sh, B->q_sh};
        ptv2_wgrad_defer_arm(kq_kept);
        const int krc = linear_wgrad_multi_hip_launcher(n, g, c, 2, gys, xs, dws, nullptr, xsc, xsh, W.kq_part, W.kq_bytes, stream);
        ptv2_wgrad_defer_arm(false);
        RUN(krc);
    }
    }
    {
        const float *gys[2] = {W.gkW, W.gqW};
        float *gxs[2] = {G->gk, G->gq};
        // (+ the reduce records of the linear_q / linear_k BatchNorm backward when the Block runtime asked for them: dense.hip)
        RUN(skinny_backward_pair_bn_reduce(n, c, g, gys, B->Ww1, gxs, stream));
    }
    ptv2_rider_flush(st);  // anything still queued (paths without a carrying launch) before the glue reads the sums
    riders.release();
    // 7. M / cW glue: finishes grad Ww1, adds the logits-path parts of grad Wp2 / bp2, grad bw1; in the same launch the
    //    folded BN_p backward (both stages contribute to the gradient of (a, b))
    {
        FoldMBwdArgs F{};
        F.c = c; F.g = g;
        F.mblocks = divup((long long)g * c, c >= 256 ? TPB : WPB) + divup((long long)c * c + c + g, TPB);
        F.blocks = F.mblocks + divup(c, TPB);
        F.Wp2 = B->Wp2; F.bp2 = B->bp2; F.Ww1 = B->Ww1; F.gM = W.gM; F.gcW = W.gcW; F.gWw1_k = W.gWw1_k; F.gWw1_q = W.gWw1_q;
        F.gWw1 = G->gWw1; F.gWp2 = G->gWp2; F.gbp2 = G->gbp2; F.gbw1 = G->gbw1;
        F.P = FoldPBwdArgs{c, B->Wp1, B->bp1, B->gamma_p, B->mu, B->cov, B->run_mean_p, B->rstd_p, B->training, W.ga1, W.gb1, W.ga2,
                           W.gb2, G->gWp1, G->gbp1, G->ggamma_p, G->gbeta_p};
        if (g_fold_scratch) {
            if (!g_fold_queue) g_fold_queue = new std::vector<FoldMBwdArgs>();
            g_fold_queue->push_back(F);
        } else {
            hipLaunchKernelGGL(fold_m_bwd_kernel, dim3(F.blocks), dim3(TPB), 0, st, F);
        }
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
