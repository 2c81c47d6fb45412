// ao_amd/csrc/gva_peb.hip -- the grouped positional-bias projection of the fused GVA (gfx950).
//
//   out[n, g*I+i] = out_v[n, g*I+i] + sum_c' A[n,g,c'] * Wp2[g*I+i, c'] + bp2[g*I+i] * sw[n,g]
//
// i.e. linear_p_bias[3] applied AFTER the softmax-weighted sum over neighbours (see ao_amd/ptv2/gva.py).
// It is a batch of G thin GEMMs (N x C') x (C' x I) with I = C/G = 8 output columns each -- a shape
// rocBLAS serves poorly (measured 353 us per call at N = 120k, profiles/r01_fused_v1_*).  Here a
// workgroup streams a tile of T = 256/I points of A[g] through LDS once per group (coalesced,
// read-once: the kernel is HBM-bound on A) and each thread owns one output element.
#include <algorithm>
#include <cstdlib>

#include "gva_common.h"

namespace gva {

// One lane per output element (n, c).  A workgroup keeps the Wp2 rows of its CT output channels in LDS
// (rows padded by 4 floats) and walks tiles of TP = 256/CT points; the A row of (n, group) is read straight
// from global memory as float4 -- the I lanes of a group read identical addresses, which the memory pipeline
// serves as one request, so A is streamed exactly once.
__global__ __launch_bounds__(TPB) void peb_fwd_kernel(int n, int c, int g, int ct, const float *__restrict__ A,
                                                      const float *__restrict__ Wp2, const float *__restrict__ bp2,
                                                      const float *__restrict__ sw, const float *__restrict__ out_v,
                                                      float *__restrict__ out) {
    extern __shared__ float4 lds4[];
    float *sW = (float *)lds4;  // [ct][c + 4]
    const int ldw = c + 4;
    const int c0 = blockIdx.y * ct;
    const int cq = c >> 2;
    for (int e = threadIdx.x; e < ct * cq; e += TPB) {
        const int r = e / cq, q = e - r * cq;
        if (c0 + r < c) *(float4 *)(sW + (size_t)r * ldw + 4 * q) = *(const float4 *)(Wp2 + (size_t)(c0 + r) * c + 4 * q);
    }
    __syncthreads();
    const int I = c / g;
    const int tp = TPB / ct;
    const int p = threadIdx.x / ct, cl = threadIdx.x - p * ct;
    const int ch = c0 + cl;
    if (p >= tp || ch >= c) return;
    const int gi = ch / I;
    const float bias = bp2[ch];
    const float4 *wr = (const float4 *)(sW + (size_t)cl * ldw);
    for (long long pt = (long long)blockIdx.x * tp + p; pt < n; pt += (long long)gridDim.x * tp) {
        const float4 *ar = (const float4 *)(A + ((size_t)pt * g + gi) * c);
        // 8 (then 4) row quads are requested before the first one is consumed: with two loads in flight per lane the
        // kernel sat parked on memory for 81 % of its wave cycles (profiles/r02_final_sq_counters.jsonl)
        float acc0 = 0.f, acc1 = 0.f;
        int q = 0;
        for (; q + 8 <= cq; q += 8) {
            float4 x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = ar[q + j];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float4 w0 = wr[q + j], w1 = wr[q + j + 1];
                acc0 = __builtin_fmaf(x[j].x, w0.x, acc0); acc0 = __builtin_fmaf(x[j].y, w0.y, acc0);
                acc0 = __builtin_fmaf(x[j].z, w0.z, acc0); acc0 = __builtin_fmaf(x[j].w, w0.w, acc0);
                acc1 = __builtin_fmaf(x[j + 1].x, w1.x, acc1); acc1 = __builtin_fmaf(x[j + 1].y, w1.y, acc1);
                acc1 = __builtin_fmaf(x[j + 1].z, w1.z, acc1); acc1 = __builtin_fmaf(x[j + 1].w, w1.w, acc1);
            }
        }
        for (; q + 4 <= cq; q += 4) {
            float4 x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = ar[q + j];
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const float4 w0 = wr[q + j], w1 = wr[q + j + 1];
                acc0 = __builtin_fmaf(x[j].x, w0.x, acc0); acc0 = __builtin_fmaf(x[j].y, w0.y, acc0);
                acc0 = __builtin_fmaf(x[j].z, w0.z, acc0); acc0 = __builtin_fmaf(x[j].w, w0.w, acc0);
                acc1 = __builtin_fmaf(x[j + 1].x, w1.x, acc1); acc1 = __builtin_fmaf(x[j + 1].y, w1.y, acc1);
                acc1 = __builtin_fmaf(x[j + 1].z, w1.z, acc1); acc1 = __builtin_fmaf(x[j + 1].w, w1.w, acc1);
            }
        }
        for (; q < cq; ++q) {
            const float4 x0 = ar[q], w0 = wr[q];
            acc0 = __builtin_fmaf(x0.x, w0.x, acc0); acc0 = __builtin_fmaf(x0.y, w0.y, acc0);
            acc0 = __builtin_fmaf(x0.z, w0.z, acc0); acc0 = __builtin_fmaf(x0.w, w0.w, acc0);
        }
        const size_t o = (size_t)pt * c + ch;
        out[o] = out_v[o] + (acc0 + acc1) + bias * sw[(size_t)pt * g + gi];
    }
}

// ---- the same product on the matrix cores (I = C / G = 8, every PT-v2m2 configuration) -----------------------------------
// Per group the projection is a (points x C') x (C' x 8) product.  One lane per output element (above) makes the 8 lanes of
// a group read the SAME 16 bytes of an A row per load instruction: a wavefront's request touches 8 rows x 16 B, the kernel
// is bound by the vector memory pipeline's line lookups (40 us for the 83 MB of A at 4.5 k points, 2 TB/s from cache).
// Here a wavefront owns 16 points; the four lanes (quarters q) of a point stream its A row 64 contiguous bytes per
// instruction, the 8 Wp2 rows of the group come from LDS as the A operand of V_MFMA_F32_16X16X4_F32
// (rows 8..15 of the tile are padding), and the result tile leaves 4 consecutive outputs of a point in each lane of q < 2
// (float4 store).  The next group's A rows are in flight while the current group is on the matrix core.
// A workgroup = one 64-row block x `gpw` groups; with stats != NULL its epilogue leaves the column statistics of `out` for
// this row block (sum, sum of squares about the block mean: the records of gemm.hip's epilogue, merged by
// bn_tiles_finalize) -- norm2's statistics pass and its read of `out` disappear.
typedef float v4f_peb __attribute__((ext_vector_type(4)));
template <int CTRL>
__device__ __forceinline__ float peb_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float peb_row16_sum(float v) {  // all-reduce over the 16 lanes that share lane >> 4
    v += peb_dpp<0xB1>(v);
    v += peb_dpp<0x4E>(v);
    v += peb_dpp<0x141>(v);
    v += peb_dpp<0x140>(v);
    return v;
}

template <int C, int PEB_MAX_GPW>  // PEB_MAX_GPW: groups per workgroup (compile-time: per-group registers are arrays of it)
__global__ __launch_bounds__(TPB) void peb_fwd_mfma_kernel(int n, int g, const float *__restrict__ A,
                                                           const float *__restrict__ Wp2, const float *__restrict__ bp2,
                                                           const float *__restrict__ sw, const float *__restrict__ out_v,
                                                           float *__restrict__ out, float *__restrict__ stats) {
    constexpr int QF = C / 16;                      // float4 per lane and group (a lane owns C / 4 of the c')
    constexpr int CH = QF <= 6 ? QF : (QF % 6 == 0 ? 6 : 4);  // float4 per lane and item
    constexpr int NCH = QF / CH;                    // items per group
    constexpr int ITEMS = PEB_MAX_GPW * NCH;
    // prefetch distance in items: ~15-18 float4 per lane in flight whatever C is.  One item ahead left 3 float4 (C = 48) to
    // 12 (C = 192) in flight per lane while every workgroup of the launch -- all co-resident, in lockstep -- sat in the
    // same phase: the memory system idled through everyone's staging and epilogues (57 us for 138 MB at 120 k points)
    constexpr int DIST = (CH >= 6 ? 3 : 5) < ITEMS ? (CH >= 6 ? 3 : 5) : (ITEMS > 1 ? ITEMS - 1 : 1);
    constexpr int RING = DIST + 1;
    static_assert(QF % CH == 0, "items tile the row");
    // row pitch C + 8 floats: (C + 8) / 4 is 2 mod 4 for every C in use, which puts the 16-byte slots of the 8 rows x 2
    // quarters that one ds_read_b128 lane group reads on 16 distinct slots (C + 4: rows r and r + 1 of neighbouring quarters
    // collided, 20-38 % of this kernel's LDS cycles were replays)
    constexpr int LDW = C + 8;
    constexpr int gpw = PEB_MAX_GPW;
    extern __shared__ float4 lds4[];
    float *sW = (float *)lds4;                      // [gpw * 8][C + 4]
    float *sS = sW + (size_t)gpw * 8 * LDW;         // [4 waves][gpw * 8]   (statistics)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int g0 = blockIdx.y * gpw;
    const int ng = g - g0 < gpw ? g - g0 : gpw;     // groups of this workgroup
    const long long row0 = (long long)blockIdx.x * 64;
    // (every global load of this kernel is unconditional -- clamped index or the zero pad, common.h: as `cond ? *p : 0` each one
    // was a basic block of its own and the ring prefetch below was drained with vmcnt(0) at every stage)
    {
        constexpr int WL = (PEB_MAX_GPW * 8 * (C / 4) + TPB - 1) / TPB;
#pragma unroll
        for (int b0 = 0; b0 < WL; b0 += 8) {
            float4 w4[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (b0 + i < WL) {
                    const int e = tid + (b0 + i) * TPB, r = e / (C / 4), c4 = e - r * (C / 4);
                    w4[i] = ptv2_ld_or_zero((const float4 *)(Wp2 + ((size_t)g0 * 8 + r) * C + 4 * c4), r < ng * 8);
                }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (b0 + i < WL) {
                    const int e = tid + (b0 + i) * TPB, r = e / (C / 4), c4 = e - r * (C / 4);
                    if (r < ng * 8) *(float4 *)(sW + (size_t)r * LDW + 4 * c4) = w4[i];
                }
        }
    }
    const long long pt = row0 + wid * 16 + l15;
    const bool rv = pt < n;
    // contraction index of lane quarter q in load j: c' = 16 j + 4 q + (0..3) -- the four lanes of a point read 64
    // contiguous bytes per instruction.  (A lane walking its own quarter of the row, c' = q C/4 + 4 j, touches 64 distinct
    // 128-byte lines per instruction: 2.9 TB/s against 4.7 TB/s for this form on the same 83 MB,
    // tools/probes/read_pattern_probe.hip; the one-lane-per-output kernel above has the same 16-byte requests.)
    const float *arow = A + ((size_t)(rv ? pt : 0) * g + g0) * C + 4 * q;
    auto fetch = [&](int item, float4 (&x)[CH]) {  // item = group * NCH + chunk
        const int gl = item / NCH, ck = item - gl * NCH;
        // rows past the end read the zero pad; groups past this workgroup's last repeat group 0 (their products are dropped)
        const float *src = rv ? arow + (size_t)(gl < ng ? gl : 0) * C : ptv2_zero_pad;
#pragma unroll
        for (int j = 0; j < CH; ++j) x[j] = *(const float4 *)(src + 16 * (ck * CH + j));
    };
    float4 x[RING][CH];
#pragma unroll
    for (int i = 0; i < DIST && i < ITEMS; ++i) fetch(i, x[i % RING]);
    // epilogue operands of every group of this workgroup, requested up front
    float4 ovr[PEB_MAX_GPW], bbr[PEB_MAX_GPW];
    float swr[PEB_MAX_GPW];
#pragma unroll
    for (int t = 0; t < PEB_MAX_GPW; ++t) {
        const bool ok = t < ng && rv && q < 2;
        ovr[t] = ptv2_ld_or_zero((const float4 *)(out_v + (size_t)pt * C + (g0 + t) * 8 + 4 * q), ok);
        swr[t] = ptv2_ld_or_zero(sw + (size_t)pt * g + g0 + t, ok);
        bbr[t] = ptv2_ld_or_zero((const float4 *)(bp2 + (g0 + t) * 8 + 4 * q), ok);
    }
    __syncthreads();
    float4 val[PEB_MAX_GPW];
    v4f_peb acc = (v4f_peb){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int item = 0; item < ITEMS; ++item) {
        constexpr int dummy = 0;
        (void)dummy;
        const int gl = item / NCH, ck = item - gl * NCH;
        if (item + DIST < ITEMS) fetch(item + DIST, x[(item + DIST) % RING]);
        const float *wr = sW + (size_t)(gl * 8 + (l15 & 7)) * LDW + 4 * q + 16 * ck * CH;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float4 w4 = *(const float4 *)(wr + 16 * j);
            const float4 xv = x[item % RING][j];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, xv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, xv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, xv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, xv.w, acc, 0, 0, 0);
        }
        if (ck == NCH - 1) {  // D[i][j]: i = output 4 q + reg (q < 2 valid), j = point l15
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rv && q < 2 && gl < ng) {
                const int col = (g0 + gl) * 8 + 4 * q;
                const float4 bb = bbr[gl];
                const float4 ov = ovr[gl];
                const float s = swr[gl];
                v.x = ov.x + acc[0] + bb.x * s; v.y = ov.y + acc[1] + bb.y * s;
                v.z = ov.z + acc[2] + bb.z * s; v.w = ov.w + acc[3] + bb.w * s;
                *(float4 *)(out + (size_t)pt * C + col) = v;
            }
            val[gl] = v;
            acc = (v4f_peb){0.f, 0.f, 0.f, 0.f};
        }
    }
    if (!stats) return;
    // column statistics of this 64-row block for the ng * 8 columns [g0 * 8, ...): two passes over the register values
    const int cnt = (int)((n - row0) < 64 ? (n - row0) : 64);
    const int ncol = ng * 8;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < PEB_MAX_GPW; ++t)
        if (t < ng) {
            const float sx = peb_row16_sum(val[t].x), sy = peb_row16_sum(val[t].y), sz = peb_row16_sum(val[t].z), s_w = peb_row16_sum(val[t].w);
            if (l15 == 0 && q < 2) *(float4 *)(sS + wid * ncol + t * 8 + 4 * q) = make_float4(sx, sy, sz, s_w);
        }
    __syncthreads();
    float4 mean[PEB_MAX_GPW];
#pragma unroll
    for (int t = 0; t < PEB_MAX_GPW; ++t)
        if (t < ng) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < 2) {
                a = *(const float4 *)(sS + t * 8 + 4 * q);
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float4 o = *(const float4 *)(sS + w * ncol + t * 8 + 4 * q);
                    a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
                }
            }
            mean[t] = a;  // block sums for now
        }
    __syncthreads();
    const float inv = 1.0f / (float)cnt;
    float *rec = stats + (size_t)blockIdx.x * 2 * C + (size_t)g0 * 8;
#pragma unroll
    for (int t = 0; t < PEB_MAX_GPW; ++t)
        if (t < ng) {
            if (wid == 0 && l15 == 0 && q < 2) *(float4 *)(rec + t * 8 + 4 * q) = mean[t];
            const bool ok = rv && q < 2;
            const float dx = ok ? val[t].x - mean[t].x * inv : 0.f, dy = ok ? val[t].y - mean[t].y * inv : 0.f;
            const float dz = ok ? val[t].z - mean[t].z * inv : 0.f, dw = ok ? val[t].w - mean[t].w * inv : 0.f;
            const float qx = peb_row16_sum(dx * dx), qy = peb_row16_sum(dy * dy), qz = peb_row16_sum(dz * dz), qw = peb_row16_sum(dw * dw);
            if (l15 == 0 && q < 2) *(float4 *)(sS + wid * ncol + t * 8 + 4 * q) = make_float4(qx, qy, qz, qw);
        }
    __syncthreads();
    if (wid == 0 && l15 == 0 && q < 2) {
#pragma unroll
        for (int t = 0; t < PEB_MAX_GPW; ++t)
            if (t < ng) {
                float4 a = *(const float4 *)(sS + t * 8 + 4 * q);
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float4 o = *(const float4 *)(sS + w * ncol + t * 8 + 4 * q);
                    a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
                }
                *(float4 *)(rec + C + t * 8 + 4 * q) = a;
            }
    }
}

// gA[n,g,c'] = sum_i gO[n, g*I+i] * Wp2[g*I+i, c'] ;  g_sw[n,g] = sum_i gO[n, g*I+i] * bp2[g*I+i]
// A thread owns one (group, float4 of c') for all of its points -- the launcher makes the thread count a multiple of
// g * c / 4 -- so its I rows of Wp2 stay in registers; re-reading them per output (8 x the bytes written, from L2) held
// the kernel at 2 TB/s of writes.
template <int I>
__global__ __launch_bounds__(TPB) void peb_bwd_kernel(int n, int c, int g, const float *__restrict__ gO,
                                                      const float *__restrict__ Wp2, const float *__restrict__ bp2,
                                                      float *__restrict__ gA, float *__restrict__ g_sw) {
    const int cq = c / 4, per = g * cq;  // float4 outputs per point
    const long long threads = (long long)gridDim.x * TPB, t = (long long)blockIdx.x * TPB + threadIdx.x;
    const int gq = (int)(t % per), gi = gq / cq, q = gq - gi * cq;
    float4 w[I];
    float bp[I];
#pragma unroll
    for (int i = 0; i < I; ++i) {
        w[i] = *(const float4 *)(Wp2 + (size_t)(gi * I + i) * c + 4 * q);
        bp[i] = bp2[gi * I + i];
    }
    const long long step = threads / per;
    for (long long pt = t / per; pt < n; pt += step) {
        const float *go = gO + (size_t)pt * c + gi * I;
        float s[I];
        if (I % 4 == 0) {
#pragma unroll
            for (int i = 0; i < I; i += 4) {
                const float4 v = *(const float4 *)(go + i);
                s[i] = v.x; s[i + 1] = v.y; s[i + 2] = v.z; s[i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < I; ++i) s[i] = go[i];
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < I; ++i) {
            acc.x = __builtin_fmaf(s[i], w[i].x, acc.x);
            acc.y = __builtin_fmaf(s[i], w[i].y, acc.y);
            acc.z = __builtin_fmaf(s[i], w[i].z, acc.z);
            acc.w = __builtin_fmaf(s[i], w[i].w, acc.w);
        }
        ((float4 *)gA)[(size_t)pt * per + gq] = acc;
        if (q == 0) {
            float tt = 0.f;
#pragma unroll
            for (int i = 0; i < I; ++i) tt = __builtin_fmaf(s[i], bp[i], tt);
            g_sw[(size_t)pt * g + gi] = tt;
        }
    }
}

// the same for shapes whose g * c / 4 shares too few factors with the workgroup size for the fixed assignment above
template <int I>
__global__ __launch_bounds__(TPB) void peb_bwd_any_kernel(int n, int c, int g, const float *__restrict__ gO,
                                                          const float *__restrict__ Wp2, const float *__restrict__ bp2,
                                                          float *__restrict__ gA, float *__restrict__ g_sw) {
    const int cq = c / 4;
    const long long total = (long long)g * n * cq;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const int q = (int)(e % cq);
        const long long ng = e / cq;
        const int gi = (int)(ng % g);
        const long long pt = ng / g;
        const float *go = gO + (size_t)pt * c + gi * I;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float tt = 0.f;
#pragma unroll
        for (int i = 0; i < I; ++i) {
            const float sv = go[i];
            const float4 w = *(const float4 *)(Wp2 + (size_t)(gi * I + i) * c + 4 * q);
            acc.x = __builtin_fmaf(sv, w.x, acc.x); acc.y = __builtin_fmaf(sv, w.y, acc.y);
            acc.z = __builtin_fmaf(sv, w.z, acc.z); acc.w = __builtin_fmaf(sv, w.w, acc.w);
            tt = __builtin_fmaf(sv, bp2[gi * I + i], tt);
        }
        ((float4 *)gA)[e] = acc;
        if (q == 0) g_sw[(size_t)pt * g + gi] = tt;
    }
}

}  // namespace gva

using namespace gva;

#define PEB_DISPATCH_I(i, CALL)        \
    switch (i) {                       \
        case 2: { CALL(2); break; }    \
        case 4: { CALL(4); break; }    \
        case 8: { CALL(8); break; }    \
        case 16: { CALL(16); break; }  \
        default: return PTV2_ERR_ARG;  \
    }

// internal (gva_block.hip): stats != NULL asks for the per-64-row-block column statistics of `out` (bn_tiles_floats(n, c)
// floats; see peb_fwd_mfma_kernel); *stats_done tells whether this call produced them (the matrix-core form only)
int gva_peb_forward_stats(int n, int c, int g, const float *A, const float *Wp2, const float *bp2, const float *sw,
                          const float *out_v, float *out, float *stats, int *stats_done, void *stream);

extern "C" int gva_peb_forward_hip_launcher(int n, int c, int g, const float *A, const float *Wp2, const float *bp2,
                                            const float *sw, const float *out_v, float *out, void *stream) {
    return gva_peb_forward_stats(n, c, g, A, Wp2, bp2, sw, out_v, out, nullptr, nullptr, stream);
}

template <int C, int GPW>
static void launch_peb_mfma_g(int n, int g, const float *A, const float *Wp2, const float *bp2, const float *sw, const float *out_v,
                              float *out, float *stats, hipStream_t st) {
    const int nrb = (n + 63) / 64;
    const size_t lds = sizeof(float) * ((size_t)GPW * 8 * (C + 8) + 4 * GPW * 8);
    auto kern = peb_fwd_mfma_kernel<C, GPW>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(nrb, (g + GPW - 1) / GPW), dim3(TPB), lds, st, n, g, A, Wp2, bp2, sw, out_v, out, stats);
}

template <int C>
static void launch_peb_mfma(int n, int g, const float *A, const float *Wp2, const float *bp2, const float *sw, const float *out_v,
                            float *out, float *stats, hipStream_t st) {
    const int nrb = (n + 63) / 64;
    // groups per workgroup: as many (of 6, 3, 2, 1) as keep >= ~512 workgroups in the launch (each stages its Wp2 rows once)
    const int opts[4] = {6, 3, 2, 1};
    int gpw = 1;
    for (int i = 0; i < 4; ++i)
        if ((long long)nrb * ((g + opts[i] - 1) / opts[i]) >= 512 || opts[i] == 1) { gpw = opts[i]; break; }
    switch (gpw) {
        case 6: launch_peb_mfma_g<C, 6>(n, g, A, Wp2, bp2, sw, out_v, out, stats, st); break;
        case 3: launch_peb_mfma_g<C, 3>(n, g, A, Wp2, bp2, sw, out_v, out, stats, st); break;
        case 2: launch_peb_mfma_g<C, 2>(n, g, A, Wp2, bp2, sw, out_v, out, stats, st); break;
        default: launch_peb_mfma_g<C, 1>(n, g, A, Wp2, bp2, sw, out_v, out, stats, st); break;
    }
}

int gva_peb_forward_stats(int n, int c, int g, const float *A, const float *Wp2, const float *bp2, const float *sw,
                          const float *out_v, float *out, float *stats, int *stats_done, void *stream) {
    if (n < 0 || c < 4 || g < 1 || c % g != 0 || c % 4 != 0) return PTV2_ERR_ARG;
    if (stats_done) *stats_done = 0;
    if (n == 0) return PTV2_OK;
    if (c / g == 8 && (c == 48 || c == 96 || c == 192 || c == 384 || c == 512)) {
        hipStream_t st = (hipStream_t)stream;
        PtvScopedTimer t(KID_PEB_FWD, st, 4.0 * ((double)n * g * c + 2.0 * n * c + (double)n * g + (double)c * c));
        switch (c) {
            case 48: launch_peb_mfma<48>(n, g, A, Wp2, bp2, sw, out_v, out, stats, st); break;
            case 96: launch_peb_mfma<96>(n, g, A, Wp2, bp2, sw, out_v, out, stats, st); break;
            case 192: launch_peb_mfma<192>(n, g, A, Wp2, bp2, sw, out_v, out, stats, st); break;
            case 384: launch_peb_mfma<384>(n, g, A, Wp2, bp2, sw, out_v, out, stats, st); break;
            default: launch_peb_mfma<512>(n, g, A, Wp2, bp2, sw, out_v, out, stats, st); break;
        }
        if (stats && stats_done) *stats_done = 64;
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    // output channels per workgroup.  Wide C: 32 (not 64) -- the 64-channel stage of Wp2 rows is 99 KB of LDS at C = 384,
    // one workgroup per CU and the kernel parked on its A-row loads; 32 channels leave room for three (58 -> 37 us at
    // C = 384, 34 -> 21 us at C = 512, unchanged at C = 192)
    const int ct = c <= 128 ? c : 32;
    if (ct > TPB) return PTV2_ERR_ARG;
    const size_t lds = sizeof(float) * (size_t)ct * (c + 4);
    if (lds > 160 * 1024) return PTV2_ERR_ARG;
    const int tp = TPB / ct;
    // every workgroup first stages its ct rows of Wp2 (up to 99 KB) in LDS: keep the whole grid co-resident, so that this
    // is paid once per CU slot and not once per round (C = 384: 6 column blocks x 256 workgroups at one per CU was six
    // rounds of staging, 106 us for 80 MB)
    const int ny = (c + ct - 1) / ct;
    const long long slots = 256LL * std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds, 1));
    const long long cap = std::min<long long>(lds > 40 * 1024 ? 256 : 1024, std::max<long long>(1, slots / ny));
    const int gx = (int)std::min<long long>(((long long)n + tp - 1) / tp, cap);
    if (lds > 32 * 1024)
        (void)hipFuncSetAttribute((const void *)peb_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    {
        PtvScopedTimer t(KID_PEB_FWD, (hipStream_t)stream, 4.0 * ((double)n * g * c + 2.0 * n * c + (double)n * g + (double)c * c));
        hipLaunchKernelGGL(peb_fwd_kernel, dim3(gx, ny), dim3(TPB), lds, (hipStream_t)stream, n, c, g, ct, A, Wp2,
                           bp2, sw, out_v, out);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int gva_peb_backward_hip_launcher(int n, int c, int g, const float *g_out, const float *Wp2,
                                             const float *bp2, float *g_A, float *g_sw, void *stream) {
    if (n < 0 || c < 4 || g < 1 || c % g != 0 || c % 4 != 0) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const int I = c / g;
    const long long total = (long long)g * n * (c / 4);
    // thread count = a multiple of the float4 outputs per point (so that a thread keeps its (group, c') for every point)
    const long long per = (long long)g * (c / 4);
    long long a = per, b = TPB;
    while (b) { const long long r = a % b; a = b; b = r; }
    const long long unit = per / a;  // workgroups per whole number of points
    const long long want = std::min<long long>((total + TPB - 1) / TPB, 256 * 16);
    const bool fixed = unit <= 256 * 4;
    const int nblk = fixed ? (int)std::max<long long>(unit, want / unit * unit) : (int)want;
#define CALL(II)                                                                                                          \
    if (fixed)                                                                                                            \
        hipLaunchKernelGGL(peb_bwd_kernel<II>, dim3(nblk), dim3(TPB), 0, (hipStream_t)stream, n, c, g, g_out, Wp2, bp2, g_A, g_sw); \
    else                                                                                                                  \
        hipLaunchKernelGGL(peb_bwd_any_kernel<II>, dim3(nblk), dim3(TPB), 0, (hipStream_t)stream, n, c, g, g_out, Wp2, bp2, g_A, g_sw)
    {
        PtvScopedTimer t(KID_PEB_BWD, (hipStream_t)stream, 4.0 * ((double)n * g * c + (double)n * c + (double)n * g + (double)c * c));
        PEB_DISPATCH_I(I, CALL)
    }
#undef CALL
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
