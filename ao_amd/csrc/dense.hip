// ao_amd/csrc/dense.hip -- the per-point (N,C) layers around the attention on gfx950:
//   * BatchNorm1d over N rows (training / eval) with optional fused ReLU, forward and backward,
//   * the weight gradient of nn.Linear as a split-K reduction.
//
// Why these exist.  PT-v2m2 wraps every Linear in PointBatchNorm (+ReLU)
// (point_transformer_v2m2_base.py:26-45,67-76,153-177): 86 BatchNorms per training step at S3DIS sizes.
// The stock channels-last BN kernels stream (N,48..384) fp32 at ~0.65 TB/s (profiles/r01_fused_v1_*),
// and the weight gradients dW = dY^T X have a 48x48 .. 384x384 output with K = N up to 1.2e5, for which
// the BLAS picks a 9-workgroup kernel (353 us per call).  Both are pure HBM streaming problems:
//   bn_stats / bn_backward_reduce  column sums over row chunks, float4 per lane, per-block partials +
//                                  fixed-order final (bitwise reproducible); the finalizer also writes
//                                  mean / rstd and updates the running statistics in place
//   bn_apply / bn_backward_apply   one read-modify-write pass each
//   linear_wgrad                   each workgroup owns a 48x48 output tile for a chunk of rows (operands
//                                  staged through LDS, 3x3 register patch per lane), partial tiles
//                                  summed in fixed order; the bias gradient falls out of the same pass
#include <algorithm>
#include <vector>
#include <cstdlib>

#include "gva_common.h"
#include "wgrad_job.h"

namespace dense {

using gva::finalize_kernel;
using gva::launch_finalize;
constexpr int TPB = 256;
constexpr int MAX_BLK = 512;

// ------------------------------------------------------------------ BN: stats --
// lanes: (row lane, float4 column quad); requires c % 4 == 0
__global__ __launch_bounds__(TPB) void bn_stats_kernel(int n, int c, const float *__restrict__ x,
                                                       float *__restrict__ part) {
    extern __shared__ float4 lds4[];
    const int cq = c >> 2;
    const int rl = TPB / cq;                 // row lanes per block (>= 1 for c <= 1024)
    const int q = threadIdx.x % cq, r = threadIdx.x / cq;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    // sums of (x - x[0,:]): shifting by one sample of the column removes the catastrophic cancellation of
    // E[x^2] - E[x]^2 when |mean| >> std, at no extra pass
    const float4 sft = ((const float4 *)x)[q];
    if (r < rl)
#pragma unroll 4
        for (long long row = (long long)blockIdx.x * rl + r; row < n; row += (long long)gridDim.x * rl) {
            float4 v = ((const float4 *)x)[row * cq + q];
            v.x -= sft.x; v.y -= sft.y; v.z -= sft.z; v.w -= sft.w;
            s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
            s2.x = __builtin_fmaf(v.x, v.x, s2.x); s2.y = __builtin_fmaf(v.y, v.y, s2.y);
            s2.z = __builtin_fmaf(v.z, v.z, s2.z); s2.w = __builtin_fmaf(v.w, v.w, s2.w);
        }
    float4 *sa = lds4, *sb = lds4 + TPB;
    sa[threadIdx.x] = s1;
    sb[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < cq) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        for (int k = 0; k < rl; ++k) {
            const float4 u = sa[k * cq + threadIdx.x], w = sb[k * cq + threadIdx.x];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
        }
        float *p = part + (size_t)blockIdx.x * 2 * c;
        ((float4 *)p)[threadIdx.x] = a;
        ((float4 *)(p + c))[threadIdx.x] = b;
    }
}

// finalize: column sums of (x-x0) and (x-x0)^2 over the per-block partials -> mean, rstd, running statistics
template <int COLS>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float *__restrict__ part, int nblk, int c, int n, const float *__restrict__ x0, float eps, float momentum,
    float *__restrict__ mean, float *__restrict__ rstd, float *run_mean, float *run_var, long long *batches,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ sc, float *__restrict__ sh) {
    constexpr int SLICES = 1024 / COLS;
    __shared__ double s1[SLICES][COLS], s2[SLICES][COLS];
    const int col = threadIdx.x & (COLS - 1), sl = threadIdx.x / COLS;
    const int ch = blockIdx.x * COLS + col;
    double a = 0.0, b = 0.0, a2 = 0.0, b2 = 0.0;
    if (ch < c) {
        int k = sl;
        for (; k + SLICES < nblk; k += 2 * SLICES) {
            a += (double)part[(size_t)k * 2 * c + ch];
            b += (double)part[(size_t)k * 2 * c + c + ch];
            a2 += (double)part[(size_t)(k + SLICES) * 2 * c + ch];
            b2 += (double)part[(size_t)(k + SLICES) * 2 * c + c + ch];
        }
        for (; k < nblk; k += SLICES) {
            a += (double)part[(size_t)k * 2 * c + ch];
            b += (double)part[(size_t)k * 2 * c + c + ch];
        }
    }
    s1[sl][col] = a + a2;
    s2[sl][col] = b + b2;
    __syncthreads();
    if (sl == 0 && ch < c) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll 8
        for (int t = 0; t < SLICES; ++t) { t1 += s1[t][col]; t2 += s2[t][col]; }
        const double d = t1 / n;                       // mean of the shifted samples
        const double m = (double)x0[ch] + d;
        double var = t2 / n - d * d;
        var = var > 0.0 ? var : 0.0;
        mean[ch] = (float)m;
        rstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
        if (sc) {  // y = x * sc + sh is the whole normalisation: consumers apply it on their operand load
            const float scale = rstd[ch] * gamma[ch];
            sc[ch] = scale;
            sh[ch] = beta[ch] - mean[ch] * scale;
        }
        if (run_mean) {
            const double unb = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
            run_mean[ch] = (float)((1.0 - momentum) * (double)run_mean[ch] + momentum * m);
            run_var[ch] = (float)((1.0 - momentum) * (double)run_var[ch] + momentum * unb);
            if (ch == 0 && batches) *batches += 1;
        }
    }
}

// the same from the row GEMM's epilogue records part[nrb][2][c] (per 64-row block: column sums and sums of squares
// about the block mean), merged with the parallel-variance identity  M2 = sum_b (M2_b + S_b^2 / n_b) - n mean^2
// one or two tensors per launch (blockIdx.z selects the set: the q / k BatchNorms of a Block are finished together)
struct BnTileSet {
    const float *part;
    float *mean, *rstd, *run_mean, *run_var;
    long long *batches;
    const float *gamma, *beta;
    float *sc, *sh;
    double *fold;  // two-level scratch (bn_fold_tiles_kernel)
    int rb;        // rows per record: 64 (the row GEMM's epilogue, the projection kernels) or 16 (gva_fwd_tile.hip); 0 = 64
};
__device__ __forceinline__ int bn_tile_rows(const BnTileSet &S, int k, int n) {  // rows of record k
    const int rb = S.rb ? S.rb : 64;
    return (n - k * rb) < rb ? (n - k * rb) : rb;
}

__device__ __forceinline__ void bn_tiles_emit(const BnTileSet &S, int ch, double t1, double t2, int n, float eps, float momentum) {
    const double m = t1 / n;
    double var = t2 / n - m * m;
    var = var > 0.0 ? var : 0.0;
    S.mean[ch] = (float)m;
    S.rstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (S.sc) {
        const float scale = S.rstd[ch] * S.gamma[ch];
        S.sc[ch] = scale;
        S.sh[ch] = S.beta[ch] - S.mean[ch] * scale;
    }
    if (S.run_mean) {
        const double unb = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
        S.run_mean[ch] = (float)((1.0 - momentum) * (double)S.run_mean[ch] + momentum * m);
        S.run_var[ch] = (float)((1.0 - momentum) * (double)S.run_var[ch] + momentum * unb);
        if (ch == 0 && S.batches) *S.batches += 1;
    }
}

template <int COLS>
__global__ __launch_bounds__(1024) void bn_finalize_tiles_kernel(BnTileSet A, BnTileSet B, int nrb, int c, int n, float eps,
                                                                 float momentum) {
    constexpr int SLICES = 1024 / COLS;
    __shared__ double s1[SLICES][COLS], s2[SLICES][COLS];
    const BnTileSet &S = blockIdx.z ? B : A;
    const float *__restrict__ part = S.part;
    const int col = threadIdx.x & (COLS - 1), sl = threadIdx.x / COLS;
    const int ch = blockIdx.x * COLS + col;
    double a = 0.0, b = 0.0, a2 = 0.0, b2 = 0.0;
    if (ch < c) {
        auto rec = [&](int k, double &sa, double &sq) {
            const int cnt = bn_tile_rows(S, k, n);
            const double sb = (double)part[(size_t)k * 2 * c + ch];
            sa += sb;
            sq += (double)part[(size_t)k * 2 * c + c + ch] + sb * sb / (double)cnt;
        };
        int k = sl;
        // eight records (16 loads) in flight per trip, added in the order of the two-chain loop below (same bits): at the full
        // resolution (1 875 records, 3-6 workgroups) that loop was 15 dependent trips, 12.5 us on the critical path of every
        // BatchNorm of a level-0 Block
        for (; k + 7 * SLICES < nrb; k += 8 * SLICES) {
            float s[8], m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s[u] = part[(size_t)(k + u * SLICES) * 2 * c + ch];
                m[u] = part[(size_t)(k + u * SLICES) * 2 * c + c + ch];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k + u * SLICES;
                const int cnt = bn_tile_rows(S, kk, n);
                const double sb = (double)s[u];
                if (u & 1) { a2 += sb; b2 += (double)m[u] + sb * sb / (double)cnt; }
                else { a += sb; b += (double)m[u] + sb * sb / (double)cnt; }
            }
        }
        for (; k + SLICES < nrb; k += 2 * SLICES) {  // two independent chains: the loads of both records are in flight
            rec(k, a, b);
            rec(k + SLICES, a2, b2);
        }
        for (; k < nrb; k += SLICES) rec(k, a, b);
    }
    s1[sl][col] = a + a2;
    s2[sl][col] = b + b2;
    __syncthreads();
    if (sl == 0 && ch < c) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll 8
        for (int t = 0; t < SLICES; ++t) { t1 += s1[t][col]; t2 += s2[t][col]; }
        bn_tiles_emit(S, ch, t1, t2, n, eps, momentum);
    }
}

// the same for MANY records (the full-resolution level: 1 875 records, c / 16 = 3 column blocks): 3 workgroups walking 29
// records per thread were 12 us of dependent round trips on the critical path of every BatchNorm of a level-0 Block.  Here
// NS workgroups per column block each fold a share of the records into one float64 partial (S.fold), and the last of them to
// arrive (one counter per column block and tensor; the workgroups are few and the partials 256 bytes, so the arrival protocol
// is cheap here) adds the NS partials in index order and emits.
constexpr int BNT_NS = 8;
__global__ __launch_bounds__(1024) void bn_finalize_tiles_split_kernel(BnTileSet A, BnTileSet B, int nrb, int c, int n, float eps,
                                                                       float momentum, unsigned *counters) {
    constexpr int COLS = 16, SLICES = 64;
    __shared__ double s1[SLICES][COLS], s2[SLICES][COLS];
    __shared__ int s_last;
    const BnTileSet &S = blockIdx.z ? B : A;
    const float *__restrict__ part = S.part;
    const int col = threadIdx.x & (COLS - 1), sl = threadIdx.x / COLS;
    const int ch = blockIdx.x * COLS + col;
    double a = 0.0, b = 0.0;
    if (ch < c) {
        for (int k = blockIdx.y * SLICES + sl; k < nrb; k += BNT_NS * SLICES) {
            const int cnt = bn_tile_rows(S, k, n);
            const double sb = (double)part[(size_t)k * 2 * c + ch];
            a += sb;
            b += (double)part[(size_t)k * 2 * c + c + ch] + sb * sb / (double)cnt;
        }
    }
    s1[sl][col] = a;
    s2[sl][col] = b;
    __syncthreads();
    double *fold = S.fold + ((size_t)blockIdx.y * 2) * c;  // [NS][2][c]
    if (sl == 0 && ch < c) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll 8
        for (int t = 0; t < SLICES; ++t) { t1 += s1[t][col]; t2 += s2[t][col]; }
        __hip_atomic_store(fold + ch, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(fold + c + ch, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the fold stores are acknowledged before the arrival is published (gva_common.h: last_block_arrives)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *cnt = counters + blockIdx.z * gridDim.x + blockIdx.x;
        const unsigned prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == BNT_NS - 1;
        if (s_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (sl == 0 && ch < c) {
        double t1 = 0.0, t2 = 0.0;
        for (int p = 0; p < BNT_NS; ++p) {
            t1 += __hip_atomic_load(S.fold + ((size_t)p * 2) * c + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t2 += __hip_atomic_load(S.fold + ((size_t)p * 2) * c + c + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bn_tiles_emit(S, ch, t1, t2, n, eps, momentum);
    }
}

// ------------------------------------------------------------------ BN: apply --
__global__ __launch_bounds__(TPB) void bn_apply_kernel(long long total4, int cq, const float *__restrict__ x,
                                                       const float *__restrict__ mean, const float *__restrict__ rstd,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       int relu, float *__restrict__ y) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total4; e += (long long)gridDim.x * TPB) {
        const int q = (int)(e % cq);
        const float4 v = ((const float4 *)x)[e];
        const float4 m = ((const float4 *)mean)[q], r = ((const float4 *)rstd)[q];
        const float4 g = ((const float4 *)gamma)[q], b = ((const float4 *)beta)[q];
        float4 o;
        o.x = __builtin_fmaf((v.x - m.x) * r.x, g.x, b.x);
        o.y = __builtin_fmaf((v.y - m.y) * r.y, g.y, b.y);
        o.z = __builtin_fmaf((v.z - m.z) * r.z, g.z, b.z);
        o.w = __builtin_fmaf((v.w - m.w) * r.w, g.w, b.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        ((float4 *)y)[e] = o;
    }
}

// Block tail fused into the last BatchNorm of a Block (point_transformer_v2m2_base.py:174-176):
//   y = ReLU(residual + rowscale[n] * BN(x))      rowscale = per-point DropPath factor (0 or 1/keep), may be NULL
__global__ __launch_bounds__(TPB) void bn_apply_residual_kernel(long long total4, int cq, const float *__restrict__ x,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ rstd,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta,
                                                                const float *__restrict__ residual,
                                                                const float *__restrict__ rowscale,
                                                                float *__restrict__ y) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total4; e += (long long)gridDim.x * TPB) {
        const int q = (int)(e % cq);
        const float rsc = rowscale ? rowscale[e / cq] : 1.f;
        const float4 v = ((const float4 *)x)[e], res = ((const float4 *)residual)[e];
        const float4 m = ((const float4 *)mean)[q], r = ((const float4 *)rstd)[q];
        const float4 g = ((const float4 *)gamma)[q], b = ((const float4 *)beta)[q];
        float4 o;
        o.x = fmaxf(__builtin_fmaf(rsc, __builtin_fmaf((v.x - m.x) * r.x, g.x, b.x), res.x), 0.f);
        o.y = fmaxf(__builtin_fmaf(rsc, __builtin_fmaf((v.y - m.y) * r.y, g.y, b.y), res.y), 0.f);
        o.z = fmaxf(__builtin_fmaf(rsc, __builtin_fmaf((v.z - m.z) * r.z, g.z, b.z), res.z), 0.f);
        o.w = fmaxf(__builtin_fmaf(rsc, __builtin_fmaf((v.w - m.w) * r.w, g.w, b.w), res.w), 0.f);
        ((float4 *)y)[e] = o;
    }
}

// backward of the fused tail: d = gy * (y > 0) is the gradient of the residual; d * rowscale[n] enters the BN backward
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_residual_kernel(int n, int c, const float *__restrict__ x,
                                                                     const float *__restrict__ gy,
                                                                     const float *__restrict__ y,
                                                                     const float *__restrict__ rowscale,
                                                                     const float *__restrict__ mean,
                                                                     const float *__restrict__ rstd,
                                                                     float *__restrict__ part) {
    extern __shared__ float4 lds4[];
    const int cq = c >> 2;
    const int rl = TPB / cq;
    const int q = threadIdx.x % cq, r = threadIdx.x / cq;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (r < rl) {
        const float4 m = ((const float4 *)mean)[q], rs = ((const float4 *)rstd)[q];
#pragma unroll 4
        for (long long row = (long long)blockIdx.x * rl + r; row < n; row += (long long)gridDim.x * rl) {
            const float4 v = ((const float4 *)x)[row * cq + q], o = ((const float4 *)y)[row * cq + q];
            float4 d = ((const float4 *)gy)[row * cq + q];
            const float rsc = rowscale ? rowscale[row] : 1.f;
            d.x = o.x > 0.f ? d.x * rsc : 0.f; d.y = o.y > 0.f ? d.y * rsc : 0.f;
            d.z = o.z > 0.f ? d.z * rsc : 0.f; d.w = o.w > 0.f ? d.w * rsc : 0.f;
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            s2.x = __builtin_fmaf(d.x, (v.x - m.x) * rs.x, s2.x); s2.y = __builtin_fmaf(d.y, (v.y - m.y) * rs.y, s2.y);
            s2.z = __builtin_fmaf(d.z, (v.z - m.z) * rs.z, s2.z); s2.w = __builtin_fmaf(d.w, (v.w - m.w) * rs.w, s2.w);
        }
    }
    float4 *sa = lds4, *sb = lds4 + TPB;
    sa[threadIdx.x] = s1;
    sb[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < cq) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b2 = a;
        for (int k = 0; k < rl; ++k) {
            const float4 u = sa[k * cq + threadIdx.x], w = sb[k * cq + threadIdx.x];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b2.x += w.x; b2.y += w.y; b2.z += w.z; b2.w += w.w;
        }
        float *p = part + (size_t)blockIdx.x * 2 * c;
        ((float4 *)p)[threadIdx.x] = a;
        ((float4 *)(p + c))[threadIdx.x] = b2;
    }
}

__global__ __launch_bounds__(TPB) void bn_bwd_apply_residual_kernel(
    long long total4, int cq, float inv_n, const float *__restrict__ x, const float *__restrict__ gy,
    const float *__restrict__ y, const float *__restrict__ rowscale, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ dbeta,
    const float *__restrict__ dgamma, int training, float *__restrict__ gx, float *__restrict__ g_residual) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total4; e += (long long)gridDim.x * TPB) {
        const int q = (int)(e % cq);
        const float rsc = rowscale ? rowscale[e / cq] : 1.f;
        const float4 v = ((const float4 *)x)[e], o = ((const float4 *)y)[e];
        float4 d = ((const float4 *)gy)[e];
        d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f; d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
        ((float4 *)g_residual)[e] = d;
        d.x *= rsc; d.y *= rsc; d.z *= rsc; d.w *= rsc;
        const float4 m = ((const float4 *)mean)[q], rs = ((const float4 *)rstd)[q], g = ((const float4 *)gamma)[q];
        float4 out;
        if (training) {
            const float4 db = ((const float4 *)dbeta)[q], dg = ((const float4 *)dgamma)[q];
            out.x = g.x * rs.x * (d.x - db.x * inv_n - (v.x - m.x) * rs.x * dg.x * inv_n);
            out.y = g.y * rs.y * (d.y - db.y * inv_n - (v.y - m.y) * rs.y * dg.y * inv_n);
            out.z = g.z * rs.z * (d.z - db.z * inv_n - (v.z - m.z) * rs.z * dg.z * inv_n);
            out.w = g.w * rs.w * (d.w - db.w * inv_n - (v.w - m.w) * rs.w * dg.w * inv_n);
        } else {
            out.x = g.x * rs.x * d.x; out.y = g.y * rs.y * d.y; out.z = g.z * rs.z * d.z; out.w = g.w * rs.w * d.w;
        }
        ((float4 *)gx)[e] = out;
    }
}

// a second, independent BatchNorm of the same shape handled by blockIdx.y == 1 of the same launches (linear_q and
// linear_k of a Block: their backward chains are independent, batching them saves three launches per Block)
struct BnSecond {
    const float *x, *gy, *mean, *rstd, *gamma, *beta;
    float *gx, *dgamma, *dbeta;
};

// -------------------------------------------------------- BN: backward reduce --
// partial columns [0,c): sum gy' ; [c,2c): sum gy' * xhat, with gy' = gy masked by the fused ReLU
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_kernel(int n, int c, const float *x, const float *gy,
                                                            const float *mean, const float *rstd, const float *gamma,
                                                            const float *beta, int relu, float *__restrict__ part,
                                                            BnSecond second) {
    extern __shared__ float4 lds4[];
    if (blockIdx.y) { x = second.x; gy = second.gy; mean = second.mean; rstd = second.rstd; gamma = second.gamma; beta = second.beta; }
    const int cq = c >> 2;
    const int rl = TPB / cq;
    const int q = threadIdx.x % cq, r = threadIdx.x / cq;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (r < rl) {
        const float4 m = ((const float4 *)mean)[q], rs = ((const float4 *)rstd)[q];
        const float4 g = ((const float4 *)gamma)[q], b = ((const float4 *)beta)[q];
#pragma unroll 4
        for (long long row = (long long)blockIdx.x * rl + r; row < n; row += (long long)gridDim.x * rl) {
            const float4 v = ((const float4 *)x)[row * cq + q];
            float4 d = ((const float4 *)gy)[row * cq + q];
            float4 h;
            h.x = (v.x - m.x) * rs.x; h.y = (v.y - m.y) * rs.y; h.z = (v.z - m.z) * rs.z; h.w = (v.w - m.w) * rs.w;
            if (relu) {
                if (__builtin_fmaf(h.x, g.x, b.x) <= 0.f) d.x = 0.f;
                if (__builtin_fmaf(h.y, g.y, b.y) <= 0.f) d.y = 0.f;
                if (__builtin_fmaf(h.z, g.z, b.z) <= 0.f) d.z = 0.f;
                if (__builtin_fmaf(h.w, g.w, b.w) <= 0.f) d.w = 0.f;
            }
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            s2.x = __builtin_fmaf(d.x, h.x, s2.x); s2.y = __builtin_fmaf(d.y, h.y, s2.y);
            s2.z = __builtin_fmaf(d.z, h.z, s2.z); s2.w = __builtin_fmaf(d.w, h.w, s2.w);
        }
    }
    float4 *sa = lds4, *sb = lds4 + TPB;
    sa[threadIdx.x] = s1;
    sb[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < cq) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b2 = a;
        for (int k = 0; k < rl; ++k) {
            const float4 u = sa[k * cq + threadIdx.x], w = sb[k * cq + threadIdx.x];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b2.x += w.x; b2.y += w.y; b2.z += w.z; b2.w += w.w;
        }
        float *p = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 2 * c;  // record of a block: [set 0 | set 1]
        ((float4 *)p)[threadIdx.x] = a;
        ((float4 *)(p + c))[threadIdx.x] = b2;
    }
}

// The same reduce with its gradient formed in place: gy[row, :] = sg[row, :] W (the input gradient of the skinny Linear(c, G) in
// front of the logits: kW = k Ww1^T, qW = q Ww1^T) is computed, stored (the apply pass reads it) and summed by the lane that owns
// the float4 -- skinny_bwd_kernel and the q / k BatchNorms' reduce were two launches over the same (n, c) rows in every Block's
// backward.  blockIdx.y: the tensor (k, q); trailing workgroups in x: the queued parameter-gradient sums (riders, as skinny_bwd).
struct SkinnyBn {
    const float *sg;   // (n, cout) gradient of the projection's output
    float *gy;         // (n, c) its input gradient = the BatchNorm's output gradient (written)
    const float *x, *mean, *rstd, *gamma, *beta;
};
__global__ __launch_bounds__(TPB) void skinny_bn_bwd_reduce_kernel(int n, int c, int cout, SkinnyBn A, SkinnyBn B,
                                                                   const float *__restrict__ W, int relu, float *__restrict__ part,
                                                                   int main_blocks, gva::PtvRiders Rs) {
    extern __shared__ float4 lds4[];
    if ((int)blockIdx.x >= main_blocks) {
        if (blockIdx.y == 0) gva::rider_run(Rs, (int)blockIdx.x - main_blocks);
        return;
    }
    const SkinnyBn &S = blockIdx.y ? B : A;
    const int cq = c >> 2;
    const int rl = TPB / cq;
    const int q = threadIdx.x % cq, r = threadIdx.x / cq;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (r < rl) {
        const float4 m = ((const float4 *)S.mean)[q], rs = ((const float4 *)S.rstd)[q];
        const float4 g = ((const float4 *)S.gamma)[q], b = ((const float4 *)S.beta)[q];
        const float *wc = W + 4 * q;
        for (long long row = (long long)blockIdx.x * rl + r; row < n; row += (long long)main_blocks * rl) {
            const float4 v = ((const float4 *)S.x)[row * cq + q];
            const float *sg = S.sg + row * cout;
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            int o = 0;
            for (; o + 6 <= cout; o += 6) {  // (six outputs' loads in flight together, as skinny_bwd_kernel)
                float sv[6];
                float4 wv[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) { sv[u] = sg[o + u]; wv[u] = *(const float4 *)(wc + (size_t)(o + u) * c); }
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    d.x = __builtin_fmaf(sv[u], wv[u].x, d.x); d.y = __builtin_fmaf(sv[u], wv[u].y, d.y);
                    d.z = __builtin_fmaf(sv[u], wv[u].z, d.z); d.w = __builtin_fmaf(sv[u], wv[u].w, d.w);
                }
            }
            for (; o < cout; ++o) {
                const float sv = sg[o];
                const float4 w = *(const float4 *)(wc + (size_t)o * c);
                d.x = __builtin_fmaf(sv, w.x, d.x); d.y = __builtin_fmaf(sv, w.y, d.y);
                d.z = __builtin_fmaf(sv, w.z, d.z); d.w = __builtin_fmaf(sv, w.w, d.w);
            }
            ((float4 *)S.gy)[row * cq + q] = d;
            float4 h;
            h.x = (v.x - m.x) * rs.x; h.y = (v.y - m.y) * rs.y; h.z = (v.z - m.z) * rs.z; h.w = (v.w - m.w) * rs.w;
            if (relu) {
                if (__builtin_fmaf(h.x, g.x, b.x) <= 0.f) d.x = 0.f;
                if (__builtin_fmaf(h.y, g.y, b.y) <= 0.f) d.y = 0.f;
                if (__builtin_fmaf(h.z, g.z, b.z) <= 0.f) d.z = 0.f;
                if (__builtin_fmaf(h.w, g.w, b.w) <= 0.f) d.w = 0.f;
            }
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            s2.x = __builtin_fmaf(d.x, h.x, s2.x); s2.y = __builtin_fmaf(d.y, h.y, s2.y);
            s2.z = __builtin_fmaf(d.z, h.z, s2.z); s2.w = __builtin_fmaf(d.w, h.w, s2.w);
        }
    }
    float4 *sa = lds4, *sb = lds4 + TPB;
    sa[threadIdx.x] = s1;
    sb[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < cq) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b2 = a;
        for (int k = 0; k < rl; ++k) {
            const float4 u = sa[k * cq + threadIdx.x], w = sb[k * cq + threadIdx.x];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b2.x += w.x; b2.y += w.y; b2.z += w.z; b2.w += w.w;
        }
        float *p = part + ((size_t)blockIdx.x * 2 + blockIdx.y) * 2 * c;  // record of a block: [set 0 | set 1]
        ((float4 *)p)[threadIdx.x] = a;
        ((float4 *)(p + c))[threadIdx.x] = b2;
    }
}

// gx = gamma * rstd * (gy' - dbeta/n - xhat * dgamma/n)   (training);   gamma * rstd * gy' (eval)
__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(long long total4, int cq, float inv_n, const float *x,
                                                           const float *gy, const float *mean, const float *rstd,
                                                           const float *gamma, const float *beta, int relu,
                                                           const float *dbeta, const float *dgamma, int training, float *gx,
                                                           BnSecond second) {
    if (blockIdx.y) {
        x = second.x; gy = second.gy; mean = second.mean; rstd = second.rstd; gamma = second.gamma; beta = second.beta;
        dbeta = second.dbeta; dgamma = second.dgamma; gx = second.gx;
    }
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total4; e += (long long)gridDim.x * TPB) {
        const int q = (int)(e % cq);
        const float4 v = ((const float4 *)x)[e];
        float4 d = ((const float4 *)gy)[e];
        const float4 m = ((const float4 *)mean)[q], rs = ((const float4 *)rstd)[q];
        const float4 g = ((const float4 *)gamma)[q], b = ((const float4 *)beta)[q];
        float4 h;
        h.x = (v.x - m.x) * rs.x; h.y = (v.y - m.y) * rs.y; h.z = (v.z - m.z) * rs.z; h.w = (v.w - m.w) * rs.w;
        if (relu) {
            if (__builtin_fmaf(h.x, g.x, b.x) <= 0.f) d.x = 0.f;
            if (__builtin_fmaf(h.y, g.y, b.y) <= 0.f) d.y = 0.f;
            if (__builtin_fmaf(h.z, g.z, b.z) <= 0.f) d.z = 0.f;
            if (__builtin_fmaf(h.w, g.w, b.w) <= 0.f) d.w = 0.f;
        }
        float4 o;
        if (training) {
            const float4 db = ((const float4 *)dbeta)[q], dg = ((const float4 *)dgamma)[q];
            o.x = g.x * rs.x * (d.x - db.x * inv_n - h.x * dg.x * inv_n);
            o.y = g.y * rs.y * (d.y - db.y * inv_n - h.y * dg.y * inv_n);
            o.z = g.z * rs.z * (d.z - db.z * inv_n - h.z * dg.z * inv_n);
            o.w = g.w * rs.w * (d.w - db.w * inv_n - h.w * dg.w * inv_n);
        } else {
            o.x = g.x * rs.x * d.x; o.y = g.y * rs.y * d.y; o.z = g.z * rs.z * d.z; o.w = g.w * rs.w * d.w;
        }
        ((float4 *)gx)[e] = o;
    }
}

constexpr int FA_COLS = 32, FA_ROWS = 128;  // consumer-side record sums: stripe width (columns), rows per workgroup

// ------------------------------- BN forward tail: tile-record merge + residual apply in one launch --
// The Block tail y = ReLU(x + rowscale * BN3(h3)) at the deep levels: the workgroups of the apply kernel (64-column stripe x
// 128 rows) merge the stripe's tile records of the producing GEMM themselves (parallel-variance identity in float64, as
// bn_finalize_tiles_kernel) instead of waiting for a finalize launch; the row-chunk-0 workgroups deliver mean / rstd /
// folded affine / running statistics for the backward and the optimizer.
// RESIDUAL = false: y = ReLU(BN(x)) (`relu` = residual == NULL ... see the launcher) -- the Linear + BatchNorm + ReLU layers
// between the Blocks (GridPool.fc, UnpoolWithSkip.proj / proj_skip: model.hip linbn_forward)
template <int PLAIN>
__global__ __launch_bounds__(TPB) void bn_tiles_apply_residual_kernel(BnTileSet S, int nrb, int n, int c, float eps, float momentum,
                                                                      const float *__restrict__ x,
                                                                      const float *__restrict__ residual,
                                                                      const float *__restrict__ rowscale, float *__restrict__ y) {
    constexpr int SL = TPB / FA_COLS;  // record slices
    __shared__ double s_a[SL][FA_COLS], s_b[SL][FA_COLS];
    __shared__ __attribute__((aligned(16))) float s_mean[FA_COLS], s_rstd[FA_COLS];
    const int col0 = blockIdx.x * FA_COLS;
    const int ncol = (c - col0) < FA_COLS ? (c - col0) : FA_COLS;
    {
        const int cj = threadIdx.x & (FA_COLS - 1), sl = threadIdx.x / FA_COLS;
        double a = 0.0, b = 0.0;
        if (cj < ncol) {
            const float *p = S.part + col0 + cj;
            int k = sl;
            for (; k + 3 * SL < nrb; k += 4 * SL) {  // four records (eight loads) of this slice in flight
                float sv[4], mv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { sv[u] = p[(size_t)(k + u * SL) * 2 * c]; mv[u] = p[(size_t)(k + u * SL) * 2 * c + c]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = k + u * SL;
                    const int cnt = bn_tile_rows(S, kk, n);
                    const double sb = (double)sv[u];
                    a += sb;
                    b += (double)mv[u] + sb * sb / (double)cnt;
                }
            }
            for (; k < nrb; k += SL) {
                const int cnt = bn_tile_rows(S, k, n);
                const double sb = (double)p[(size_t)k * 2 * c];
                a += sb;
                b += (double)p[(size_t)k * 2 * c + c] + sb * sb / (double)cnt;
            }
        }
        s_a[sl][cj] = a;
        s_b[sl][cj] = b;
    }
    __syncthreads();
    if (threadIdx.x < FA_COLS) {
        const int cj = threadIdx.x;
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int t = 0; t < SL; ++t) { t1 += s_a[t][cj]; t2 += s_b[t][cj]; }
        const double m = t1 / n;
        double var = t2 / n - m * m;
        var = var > 0.0 ? var : 0.0;
        s_mean[cj] = (float)m;
        s_rstd[cj] = (float)(1.0 / sqrt(var + (double)eps));
        if (blockIdx.y == 0 && cj < ncol) bn_tiles_emit(S, col0 + cj, t1, t2, n, eps, momentum);
    }
    __syncthreads();
    constexpr int QW = FA_COLS / 4, RL = TPB / QW;  // column quads of the stripe x row lanes
    const int cq = c >> 2, q = threadIdx.x % QW, rl = threadIdx.x / QW;
    const int qcol = (col0 >> 2) + q;
    if (4 * q >= ncol) return;
    const float4 m = *(const float4 *)(s_mean + 4 * q), r = *(const float4 *)(s_rstd + 4 * q);
    const float4 g = ((const float4 *)S.gamma)[qcol], b = ((const float4 *)S.beta)[qcol];
    const long long r0 = (long long)blockIdx.y * FA_ROWS;
    const long long r1 = (r0 + FA_ROWS) < (long long)n ? (r0 + FA_ROWS) : (long long)n;
    for (long long row = r0 + rl; row < r1; row += RL) {
        const long long e = row * cq + qcol;
        if (PLAIN) {
            const float4 v = ((const float4 *)x)[e];
            float4 o;
            o.x = fmaxf(__builtin_fmaf((v.x - m.x) * r.x, g.x, b.x), 0.f);
            o.y = fmaxf(__builtin_fmaf((v.y - m.y) * r.y, g.y, b.y), 0.f);
            o.z = fmaxf(__builtin_fmaf((v.z - m.z) * r.z, g.z, b.z), 0.f);
            o.w = fmaxf(__builtin_fmaf((v.w - m.w) * r.w, g.w, b.w), 0.f);
            ((float4 *)y)[e] = o;
            continue;
        }
        const float rsc = rowscale ? rowscale[row] : 1.f;
        const float4 v = ((const float4 *)x)[e], res = ((const float4 *)residual)[e];
        float4 o;
        o.x = fmaxf(__builtin_fmaf(rsc, __builtin_fmaf((v.x - m.x) * r.x, g.x, b.x), res.x), 0.f);
        o.y = fmaxf(__builtin_fmaf(rsc, __builtin_fmaf((v.y - m.y) * r.y, g.y, b.y), res.y), 0.f);
        o.z = fmaxf(__builtin_fmaf(rsc, __builtin_fmaf((v.z - m.z) * r.z, g.z, b.z), res.z), 0.f);
        o.w = fmaxf(__builtin_fmaf(rsc, __builtin_fmaf((v.w - m.w) * r.w, g.w, b.w), res.w), 0.f);
        ((float4 *)y)[e] = o;
    }
}

// ---------------------------------------- BN backward: finalize + apply in one launch --
// Deep levels (n <= ~8 k rows: a few dozen reduce records).  The BatchNorm backward was reduce -> finalize -> apply, the
// last two 5 us launches each of which is almost all launch boundary.  Here the apply kernel's workgroups own a 64-column
// stripe x a chunk of rows and first sum the stripe's 2 x 64 record columns themselves (nrec records of the reduce pass or
// of the producing GEMM's epilogue; <= 256 records x 128 columns = 128 KB of L2 reads per workgroup, a ~2 us prologue that
// every workgroup runs concurrently), then apply.  The row-chunk-0 workgroups also deliver dbeta / dgamma.  Column sums:
// thread (column, slice of 2) walks its records in float64, slices combined in slice order -- fixed association, bitwise
// reproducible.  blockIdx.z selects one of two independent BatchNorms of the same shape (linear_q / linear_k).
struct BnFinApply {
    const float *part; int nrec, rec_floats, off;   // record r, set columns: part[r * rec_floats + off + (0..c-1: dbeta, c..2c-1: dgamma)]
    const float *x, *gy, *mean, *rstd, *gamma, *beta;
    float *gx, *dbeta, *dgamma;
    // residual tail (bn_backward_residual): the ReLU mask comes from y > 0, d * rowscale enters the BatchNorm, d itself is
    // the residual gradient
    const float *y, *rowscale;
    float *g_residual;
};

template <bool RESIDUAL>
__global__ __launch_bounds__(TPB) void bn_bwd_finapply_kernel(int n, int c, int relu, int training, float inv_n, BnFinApply A0,
                                                              BnFinApply A1) {
    const BnFinApply &A = blockIdx.z ? A1 : A0;
    constexpr int SL = TPB / (2 * FA_COLS);  // record slices
    __shared__ double s_part[SL][2 * FA_COLS];
    __shared__ __attribute__((aligned(16))) float s_db[FA_COLS], s_dg[FA_COLS];
    const int col0 = blockIdx.x * FA_COLS;
    const int ncol = (c - col0) < FA_COLS ? (c - col0) : FA_COLS;
    {   // column sums of this stripe: thread -> (record column j of 2 * FA_COLS, slice sl of SL)
        const int j = threadIdx.x & (2 * FA_COLS - 1), sl = threadIdx.x / (2 * FA_COLS);
        const int which = j / FA_COLS, cj = j - which * FA_COLS;  // 0: dbeta, 1: dgamma
        double acc = 0.0;
        if (cj < ncol) {
            const float *p = A.part + A.off + (size_t)which * c + col0 + cj;
            const size_t rs = (size_t)A.rec_floats;
            int r = sl;
            for (; r + 7 * SL < A.nrec; r += 8 * SL) {  // eight records of this slice in flight (one chain: fixed order)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(r + u * SL) * rs];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += (double)v[u];
            }
            for (; r < A.nrec; r += SL) acc += (double)p[(size_t)r * rs];
        }
        s_part[sl][j] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 2 * FA_COLS) {
        const int j = threadIdx.x, which = j / FA_COLS, cj = j - which * FA_COLS;
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < SL; ++u) t += s_part[u][j];
        const float v = (float)t;
        (which ? s_dg : s_db)[cj] = v;
        if (blockIdx.y == 0 && cj < ncol) (which ? A.dgamma : A.dbeta)[col0 + cj] = v;
    }
    __syncthreads();
    constexpr int QW = FA_COLS / 4, RL = TPB / QW;  // column quads of the stripe x row lanes
    const int cq = c >> 2, q = threadIdx.x % QW, rl = threadIdx.x / QW;
    const int qcol = (col0 >> 2) + q;
    if (4 * q >= ncol) return;
    const float4 m = ((const float4 *)A.mean)[qcol], rs = ((const float4 *)A.rstd)[qcol], g = ((const float4 *)A.gamma)[qcol];
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!RESIDUAL && relu) b = ((const float4 *)A.beta)[qcol];
    const float4 db = *(const float4 *)(s_db + 4 * q), dg = *(const float4 *)(s_dg + 4 * q);
    const long long r0 = (long long)blockIdx.y * FA_ROWS;
    const long long r1 = (r0 + FA_ROWS) < (long long)n ? (r0 + FA_ROWS) : (long long)n;
    for (long long row = r0 + rl; row < r1; row += RL) {
        const long long e = row * cq + qcol;
        const float4 v = ((const float4 *)A.x)[e];
        float4 d = ((const float4 *)A.gy)[e];
        float4 h;
        h.x = (v.x - m.x) * rs.x; h.y = (v.y - m.y) * rs.y; h.z = (v.z - m.z) * rs.z; h.w = (v.w - m.w) * rs.w;
        if (RESIDUAL) {
            const float4 o = ((const float4 *)A.y)[e];
            const float rsc = A.rowscale ? A.rowscale[row] : 1.f;
            d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f; d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
            ((float4 *)A.g_residual)[e] = d;
            d.x *= rsc; d.y *= rsc; d.z *= rsc; d.w *= rsc;
        } else if (relu) {
            if (__builtin_fmaf(h.x, g.x, b.x) <= 0.f) d.x = 0.f;
            if (__builtin_fmaf(h.y, g.y, b.y) <= 0.f) d.y = 0.f;
            if (__builtin_fmaf(h.z, g.z, b.z) <= 0.f) d.z = 0.f;
            if (__builtin_fmaf(h.w, g.w, b.w) <= 0.f) d.w = 0.f;
        }
        float4 o;
        if (training) {
            o.x = g.x * rs.x * (d.x - db.x * inv_n - h.x * dg.x * inv_n);
            o.y = g.y * rs.y * (d.y - db.y * inv_n - h.y * dg.y * inv_n);
            o.z = g.z * rs.z * (d.z - db.z * inv_n - h.z * dg.z * inv_n);
            o.w = g.w * rs.w * (d.w - db.w * inv_n - h.w * dg.w * inv_n);
        } else {
            o.x = g.x * rs.x * d.x; o.y = g.y * rs.y * d.y; o.z = g.z * rs.z * d.z; o.w = g.w * rs.w * d.w;
        }
        ((float4 *)A.gx)[e] = o;
    }
}

// records few enough for the consumer-side sum (and the A/B switch of the tests: AO_AMD_BN_FINAPPLY=0)
static bool finapply_ok(int n, int nrec) {
    const char *e = getenv("AO_AMD_BN_FINAPPLY");
    // (n <= 32768 -- the second level of the bench scene, 19 k rows -- measured the same step to 0.01 ms: the separate finalize +
    // apply pair stays there)
    return nrec <= 640 && n <= 16384 && !(e && e[0] == '0');  // (640: the 16-row records of the k-split GEMM at <= 10 k rows)
}

static void launch_finapply(hipStream_t st, int n, int c, int relu, int training, bool residual, int sets, const BnFinApply &A0,
                            const BnFinApply &A1) {
    const dim3 grid((unsigned)((c + FA_COLS - 1) / FA_COLS), (unsigned)((n + FA_ROWS - 1) / FA_ROWS), (unsigned)sets);
    if (residual)
        hipLaunchKernelGGL(bn_bwd_finapply_kernel<true>, grid, dim3(TPB), 0, st, n, c, relu, training, 1.0f / (float)n, A0, A1);
    else
        hipLaunchKernelGGL(bn_bwd_finapply_kernel<false>, grid, dim3(TPB), 0, st, n, c, relu, training, 1.0f / (float)n, A0, A1);
}

// --------------------------------------------------------------- Linear wgrad --
// dW[b][o][i] = sum_n gY[n*ldy + b*sy + o] * X[n*ldx + b*sx + i];  db[b][o] = sum_n gY[...]   (b < batch)
// fp32 MFMA 16x16x4 (exact f32 FMA chain): the reduction index n is the MFMA k; both operand fragments are
// read straight from global memory -- lane l of a fragment holds element [row0 + (l>>4)][col0 + (l&15)],
// i.e. four 64-byte row segments per load, no LDS staging.  A workgroup = 4 waves = one (up to) 48x48
// output tile for one chunk of rows; the waves interleave k-steps and are summed through LDS.
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int WG_MT = 3, WG_TILE = 16 * WG_MT, WG_CHUNK = 256, WG_CHUNK_MIN = 128;

// several independent products of one shape in one launch (blockIdx.z selects the operand pair)
struct WgradMulti {
    const float *gY[6], *X[6];
    float *dW[6], *db[6];
    int count;  // 0: the strided form (gY + z * sy, X + z * sx)
    const float *xsc[6], *xsh[6];  // != NULL: the X operand of pair z is ReLU(x * xsc + xsh) (fused BatchNorm + ReLU)
};

// BF16: the U = 8 k-steps of a trip (8 rows per lane and fragment) are exactly the 8-per-lane operand of
// V_MFMA_F32_16X16X32_BF16: 8 fp32 MFMAs per tile pair become one instruction on bf16-rounded operands.
// (a function of the workgroup's coordinates, like wgrad_lds_tile below: the per-call kernel and the batched kernel share it)
template <bool BF16>
__device__ __forceinline__ void wgrad_direct_tile(const int n, const int cout, const int cin, const int tiles_i,
                                                  const float *__restrict__ A, const long long ldy,
                                                  const float *__restrict__ B, const long long ldx, float *__restrict__ part,
                                                  const bool part_b, const int batch, const float *__restrict__ xs,
                                                  const float *__restrict__ xh, const int chunk, const int bx, const int by,
                                                  const int bz) {
    __shared__ float sRed[TPB / WAVE][WG_MT * WG_MT * 4 + WG_MT][WAVE + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int to = (by / tiles_i) * WG_TILE, ti = (by % tiles_i) * WG_TILE;
    const long long r0 = (long long)bx * chunk;
    const long long r1 = (r0 + chunk) < (long long)n ? (r0 + chunk) : (long long)n;
    const int lr = lane >> 4, lc = lane & 15;
    f32x4 acc[WG_MT][WG_MT];
    float bsum[WG_MT];
#pragma unroll
    for (int m = 0; m < WG_MT; ++m) {
        bsum[m] = 0.f;
#pragma unroll
        for (int t = 0; t < WG_MT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // (measured and rejected, round 3: columns 3 lc + m per lane so that a lane's three operand values of a k-step are ONE
    // 12-byte load and a row is read as 192 contiguous bytes -- 16 instead of 48 vector-memory instructions per trip: slower at
    // every shape, 34 -> 38 us at 4.5 k x 192 x 5 products, 58 -> 71 us at 120 k x 48 x 5; three 64-byte segments stay)
    bool mo[WG_MT], mi[WG_MT];
    float xsc_[WG_MT], xsh_[WG_MT];
#pragma unroll
    for (int m = 0; m < WG_MT; ++m) {
        mo[m] = to + m * 16 + lc < cout;
        mi[m] = ti + m * 16 + lc < cin;
        xsc_[m] = (xs && mi[m]) ? xs[ti + m * 16 + lc] : 1.f;
        xsh_[m] = (xs && mi[m]) ? xh[ti + m * 16 + lc] : 0.f;
    }
    // U k-steps per trip: all 6 U fragment loads are issued before the first MFMA consumes one (a step-by-step loop
    // paid one memory latency per 4 rows: 2.5 us per 100 rows of chunk, independent of the problem size)
    constexpr int U = 8;
    for (long long rb = r0 + 4 * wid; rb < r1; rb += 4 * (TPB / WAVE) * U) {
        float a[U][WG_MT], b[U][WG_MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long row = rb + (long long)u * 4 * (TPB / WAVE) + lr;
            const bool rok = row < r1;
#pragma unroll
            for (int m = 0; m < WG_MT; ++m) {
                a[u][m] = (rok && mo[m]) ? A[row * ldy + to + m * 16 + lc] : 0.f;
                b[u][m] = (rok && mi[m]) ? B[row * ldx + ti + m * 16 + lc] : 0.f;
            }
        }
        if (xs) {  // fused BatchNorm + ReLU on the X operand; rows past the end meet a == 0, so no masking is needed
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int m = 0; m < WG_MT; ++m) b[u][m] = fmaxf(__builtin_fmaf(b[u][m], xsc_[m], xsh_[m]), 0.f);
        }
        if constexpr (BF16) {
            ptv2_bf16x8 ab[WG_MT], bb[WG_MT];
#pragma unroll
            for (int m = 0; m < WG_MT; ++m) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bsum[m] += a[u][m];
                    ab[m][u] = (__bf16)a[u][m];
                    bb[m][u] = (__bf16)b[u][m];
                }
            }
#pragma unroll
            for (int m = 0; m < WG_MT; ++m)
#pragma unroll
                for (int t = 0; t < WG_MT; ++t)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[m], bb[t], acc[m][t], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int m = 0; m < WG_MT; ++m) {
                    bsum[m] += a[u][m];
#pragma unroll
                    for (int t = 0; t < WG_MT; ++t)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][m], b[u][t], acc[m][t], 0, 0, 0);
                }
        }
    }
    // combine the 4 waves (fixed order) and write the partial tile
#pragma unroll
    for (int m = 0; m < WG_MT; ++m) {
        float bs = bsum[m];
        bs += __shfl_xor(bs, 16, WAVE);
        bs += __shfl_xor(bs, 32, WAVE);
        sRed[wid][WG_MT * WG_MT * 4 + m][lane] = bs;
#pragma unroll
        for (int t = 0; t < WG_MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sRed[wid][(m * WG_MT + t) * 4 + r][lane] = acc[m][t][r];
    }
    __syncthreads();
    const size_t rec = (size_t)batch * cout * cin + (part_b ? (size_t)batch * cout : 0);
    float *p = part + (size_t)bx * rec + (size_t)bz * cout * cin;
    for (int e = threadIdx.x; e < WG_MT * WG_MT * 4 * WAVE; e += TPB) {
        const int q = e / WAVE, l = e - q * WAVE;
        const int mt = q / 4, r = q - mt * 4, m = mt / WG_MT, t = mt - m * WG_MT;
        const int o = to + m * 16 + (l >> 4) * 4 + r, i = ti + t * 16 + (l & 15);  // D: row=(lane>>4)*4+reg, col=lane&15
        if (o < cout && i < cin) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < TPB / WAVE; ++wv) v += sRed[wv][q][l];
            p[(size_t)o * cin + i] = v;
        }
    }
    if (part_b && ti == 0 && threadIdx.x < WG_TILE) {
        const int m = threadIdx.x >> 4, l = threadIdx.x & 15, o = to + threadIdx.x;
        if (o < cout) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < TPB / WAVE; ++wv) v += sRed[wv][WG_MT * WG_MT * 4 + m][l];
            part[(size_t)bx * rec + (size_t)batch * cout * cin + (size_t)bz * cout + o] = v;
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(TPB) void linear_wgrad_kernel(int n, int cout, int cin, int tiles_i,
                                                           const float *__restrict__ gY, long long ldy, long long sy,
                                                           const float *__restrict__ X, long long ldx, long long sx,
                                                           float *__restrict__ part, float *__restrict__ part_b,
                                                           int batch, WgradMulti multi, int chunk) {
    const int bz = blockIdx.z;
    wgrad_direct_tile<BF16>(n, cout, cin, tiles_i, multi.count ? multi.gY[bz] : gY + (long long)bz * sy, ldy,
                            multi.count ? multi.X[bz] : X + (long long)bz * sx, ldx, part, part_b != nullptr, batch,
                            multi.count ? multi.xsc[bz] : nullptr, multi.count ? multi.xsh[bz] : nullptr, chunk,
                            (int)blockIdx.x, (int)blockIdx.y, bz);
}

// ---- the same reduction with the operands staged through LDS (fp32 matrix cores) ---------------------------------------
// linear_wgrad_kernel reads its MFMA fragments straight from global memory: 48 four-byte loads per lane and trip (four
// 64-byte row segments per instruction), 72 MFMAs behind them; its waves sat in issue stalls for 60 % of their cycles with
// the matrix pipe 23 % busy (profiles/r02_final_sq_counters.jsonl).  Here a workgroup streams 64-row stages of both operand
// tiles (64 x 48 floats each) with 16-byte loads, every row a contiguous 192-byte run, into a double-buffered LDS image
// (row pitch 48 floats: the ds_read_b32 fragment reads of lanes (k = lane >> 4, column = lane & 15) fall on 32 distinct
// banks per half-wave); the next stage's loads are in flight in registers while the current one is on the matrix cores;
// each wavefront contracts 16 of the stage's 64 rows (4 k-steps x 9 tiles) and the four partial tiles are added through LDS
// at the end, exactly as in linear_wgrad_kernel (same records, same finalize).  The contraction order over the rows differs
// from that kernel's (wave w takes rows 16 w .. 16 w + 15 of every stage); results are bitwise reproducible run to run.
constexpr int WL_ROWS = 64;  // rows per stage
// RS != 0: the bias sums are WEIGHTED by a per-(row, product) scalar: db[b][o] = sum_n gY[n, b, o] * rowscale[n * lds_s + b]
// (the grouped projection's bias gradient, sum_n g_out[n, ch] sw[n, group(ch)], which was a kernel of its own per Block)
// (the body is a function of the workgroup's coordinates (bx: row chunk, by: output tile, bz: product) so that the same code
// serves the one-launch-per-call kernel below and the batched kernel that runs the deferred launches of a whole backward)
template <int RS>
__device__ __forceinline__ void wgrad_lds_tile(const int n, const int cout, const int cin, const int tiles_i,
                                               const float *__restrict__ A, const long long ldy,
                                               const float *__restrict__ B, const long long ldx,
                                               float *__restrict__ part, const bool part_b, const int batch,
                                               const float *__restrict__ xs, const float *__restrict__ xh, const int chunk,
                                               const float *__restrict__ rowscale, const long long lds_s, const int bx,
                                               const int by, const int bz) {
    extern __shared__ float4 wl_lds4[];
    float *sA = (float *)wl_lds4;                    // [2][WL_ROWS][WG_TILE]
    float *sB = sA + 2 * WL_ROWS * WG_TILE;           // [2][WL_ROWS][WG_TILE]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int to = (by / tiles_i) * WG_TILE, ti = (by % tiles_i) * WG_TILE;
    const long long r0 = (long long)bx * chunk;
    const long long r1 = (r0 + chunk) < (long long)n ? (r0 + chunk) : (long long)n;
    const int lr = lane >> 4, lc = lane & 15;
    f32x4 acc[WG_MT][WG_MT];
    float bsum[WG_MT];
#pragma unroll
    for (int m = 0; m < WG_MT; ++m) {
        bsum[m] = 0.f;
#pragma unroll
        for (int t = 0; t < WG_MT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float xsc_[WG_MT], xsh_[WG_MT];
#pragma unroll
    for (int m = 0; m < WG_MT; ++m) {
        const bool mi = ti + m * 16 + lc < cin;
        xsc_[m] = (xs && mi) ? xs[ti + m * 16 + lc] : 1.f;
        xsh_[m] = (xs && mi) ? xh[ti + m * 16 + lc] : 0.f;
    }
    // loader: thread -> float4 slots f = tid + 256 j (j < 3) of a 64 x 12 stage tile, for both operands
    constexpr int Q = WG_TILE / 4, SLOTS = WL_ROWS * Q / TPB;  // 12 float4 per row, 3 slots per thread
    float4 ra[SLOTS], rb[SLOTS];
    // (cout and cin are multiples of 4 here -- wgrad_lds_ok: a 16-byte piece is inside the tile or outside it -- and every load
    // is unconditional: rows past the chunk / pieces past the edge read the zero pad of common.h)
    float sn[4], sc_[4];  // (RS) row scalars of the stage in flight / of the stage on the matrix cores: rows 16 wid + 4 ks + lr
    auto fetch = [&](long long rs) {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int f = tid + TPB * j, row = f / Q, c4 = (f - row * Q) * 4;
            const long long r = rs + row;
            ra[j] = ptv2_ld_or_zero((const float4 *)(A + r * ldy + to + c4), r < r1 && to + c4 < cout);
            rb[j] = ptv2_ld_or_zero((const float4 *)(B + r * ldx + ti + c4), r < r1 && ti + c4 < cin);
        }
        if constexpr (RS != 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const long long r = rs + wid * 16 + ks * 4 + lr;
                sn[ks] = ptv2_ld_or_zero(rowscale + r * lds_s + bz, r < r1);
            }
        }
    };
    auto stash = [&](int buf) {
        if constexpr (RS != 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) sc_[ks] = sn[ks];
        }
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int f = tid + TPB * j;  // (row * Q + c4 / 4) * 4 floats = row * WG_TILE + c4: the image is the tile, row-major
            *(float4 *)(sA + (size_t)buf * WL_ROWS * WG_TILE + 4 * f) = ra[j];
            *(float4 *)(sB + (size_t)buf * WL_ROWS * WG_TILE + 4 * f) = rb[j];
        }
    };
    fetch(r0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (long long rs = r0; rs < r1; rs += WL_ROWS, buf ^= 1) {
        const bool more = rs + WL_ROWS < r1;
        if (more) fetch(rs + WL_ROWS);
        const float *pa = sA + (size_t)buf * WL_ROWS * WG_TILE + (size_t)(wid * 16 + lr) * WG_TILE + lc;
        const float *pb = sB + (size_t)buf * WL_ROWS * WG_TILE + (size_t)(wid * 16 + lr) * WG_TILE + lc;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {  // rows 16 wid + 4 ks + lr of the stage
            float a[WG_MT], b[WG_MT];
#pragma unroll
            for (int m = 0; m < WG_MT; ++m) {
                a[m] = pa[ks * 4 * WG_TILE + m * 16];
                b[m] = pb[ks * 4 * WG_TILE + m * 16];
            }
            if (xs) {  // fused BatchNorm + ReLU on the X operand; rows past the end meet a == 0
#pragma unroll
                for (int m = 0; m < WG_MT; ++m) b[m] = fmaxf(__builtin_fmaf(b[m], xsc_[m], xsh_[m]), 0.f);
            }
#pragma unroll
            for (int m = 0; m < WG_MT; ++m) {
                if constexpr (RS != 0) bsum[m] = __builtin_fmaf(a[m], sc_[ks], bsum[m]);
                else bsum[m] += a[m];
#pragma unroll
                for (int t = 0; t < WG_MT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[t], acc[m][t], 0, 0, 0);
            }
        }
        if (more) {
            stash(buf ^ 1);  // (the other buffer's readers finished before the barrier that ended the previous trip)
            __syncthreads();
        }
    }
    // combine the 4 waves (fixed order) and write the partial tile: as linear_wgrad_kernel (the stage buffers are dead)
    __syncthreads();
    float(*sRed)[WG_MT * WG_MT * 4 + WG_MT][WAVE + 1] = (float(*)[WG_MT * WG_MT * 4 + WG_MT][WAVE + 1]) wl_lds4;
#pragma unroll
    for (int m = 0; m < WG_MT; ++m) {
        float bs = bsum[m];
        bs += __shfl_xor(bs, 16, WAVE);
        bs += __shfl_xor(bs, 32, WAVE);
        sRed[wid][WG_MT * WG_MT * 4 + m][lane] = bs;
#pragma unroll
        for (int t = 0; t < WG_MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sRed[wid][(m * WG_MT + t) * 4 + r][lane] = acc[m][t][r];
    }
    __syncthreads();
    const size_t rec = (size_t)batch * cout * cin + (part_b ? (size_t)batch * cout : 0);
    float *p = part + (size_t)bx * rec + (size_t)bz * cout * cin;
    for (int e = threadIdx.x; e < WG_MT * WG_MT * 4 * WAVE; e += TPB) {
        const int q = e / WAVE, l = e - q * WAVE;
        const int mt = q / 4, r = q - mt * 4, m = mt / WG_MT, t = mt - m * WG_MT;
        const int o = to + m * 16 + (l >> 4) * 4 + r, i = ti + t * 16 + (l & 15);  // D: row=(lane>>4)*4+reg, col=lane&15
        if (o < cout && i < cin) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < TPB / WAVE; ++wv) v += sRed[wv][q][l];
            p[(size_t)o * cin + i] = v;
        }
    }
    if (part_b && ti == 0 && threadIdx.x < WG_TILE) {
        const int m = threadIdx.x >> 4, l = threadIdx.x & 15, o = to + threadIdx.x;
        if (o < cout) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < TPB / WAVE; ++wv) v += sRed[wv][WG_MT * WG_MT * 4 + m][l];
            part[(size_t)bx * rec + (size_t)batch * cout * cin + (size_t)bz * cout + o] = v;
        }
    }
}

template <int RS>
__global__ __launch_bounds__(TPB) void linear_wgrad_lds_kernel(int n, int cout, int cin, int tiles_i,
                                                               const float *__restrict__ gY, long long ldy, long long sy,
                                                               const float *__restrict__ X, long long ldx, long long sx,
                                                               float *__restrict__ part, float *__restrict__ part_b,
                                                               int batch, WgradMulti multi, int chunk,
                                                               const float *__restrict__ rowscale, long long lds_s) {
    const int bz = blockIdx.z;
    wgrad_lds_tile<RS>(n, cout, cin, tiles_i, multi.count ? multi.gY[bz] : gY + (long long)bz * sy, ldy,
                       multi.count ? multi.X[bz] : X + (long long)bz * sx, ldx, part, part_b != nullptr, batch,
                       multi.count ? multi.xsc[bz] : nullptr, multi.count ? multi.xsh[bz] : nullptr, chunk, rowscale, lds_s,
                       (int)blockIdx.x, (int)blockIdx.y, bz);
}

// ---- the deferred launches of a whole backward in one launch ------------------------------------------------------------
// A Block's weight gradients are off its critical chain (nothing reads dW before the optimizer), and at the deep levels each
// of their launches is a handful of latency-bound workgroups: 28-38 us for 3 MB of operands, a third of it spent alone on the
// GPU.  Inside ptv2_model_backward the eligible launches (this LDS-staged kernel, fp32) are not issued where they are called:
// the call files a job -- operands, shape, its slice of a record arena -- and ONE launch at the end of the backward runs them
// all, workgroup -> (job, chunk, tile, product) through a job table in device memory, followed by ONE finalize over the
// records of all jobs.  Kernels, tile order and record layout are those of the per-call launch; a filed job's row chunks are
// longer (wg_chunk: the other jobs fill the GPU), so its sums agree with the per-call launch's to ~2e-6 of the gradient's norm.
constexpr int WGRAD_PACK = 8;  // jobs per table-writer launch (by value: the kernarg block holds 4 KB)
struct WgradJobPack { WgradJob j[WGRAD_PACK]; };
static_assert(sizeof(WgradJobPack) + 16 <= 4096, "the table writer's argument block must fit the 4 KB kernarg segment");
__global__ void wgrad_jobs_write_kernel(WgradJobPack pack, int count, WgradJob *table) {
    if ((int)threadIdx.x < count) table[threadIdx.x] = pack.j[threadIdx.x];
}

template <int RS>
__global__ __launch_bounds__(TPB) void linear_wgrad_lds_kernel_jobs(const WgradJob *__restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].wg0) ++j;  // (uniform: scalar loads)
    const WgradJob &J = jobs[j];
    const int local = (int)blockIdx.x - J.wg0;
    const int bx = local % J.chunks, rest = local / J.chunks, by = rest % J.tiles, bz = rest / J.tiles;
    const bool multi = J.count > 0;
    wgrad_lds_tile<RS>(J.n, J.cout, J.cin, J.tiles_i, multi ? J.mgY[bz] : J.gY + (long long)bz * J.sy, J.ldy,
                       multi ? J.mX[bz] : J.X + (long long)bz * J.sx, J.ldx, J.part, J.has_pb != 0, J.batch,
                       multi ? J.mxsc[bz] : nullptr, multi ? J.mxsh[bz] : nullptr, J.chunk, J.rowscale, J.lds_s, bx, by, bz);
}

// ---- the grouped projection's weight gradient on the vector ALUs ---------------------------------------------------------------
// dWp2[g][i][:] = sum_n g_out[n, 8 g + i] A[n, g, :] and db[g][i] = sum_n g_out[n, 8 g + i] sw[n, g] (eight output rows per group:
// c / g = 8 in every PT-v2m2 configuration).  As a strided batch of (8, c) products on the matrix cores (linear_wgrad_lds_kernel
// <1>) every workgroup streamed a 48-column piece of ONE group's rows of A -- 192 bytes every g c 4 = 18 KB -- and five sixths of
// its 48 x 48 tile were padding: 2.5 TB/s over the 1.27 GB of A a step reads.  Here a workgroup takes a row chunk and a block
// of `gw` consecutive groups, i.e. a contiguous gw c 4-byte piece of every row of A (3-4 KB), one float4 of it per thread and
// row, eight float4 accumulators per thread; the row slots of the workgroup are added through LDS in slot order and the result
// is one chunk record of the strided form's layout ([g][8][c] weights, then [g][8] bias sums): same finalize.
constexpr int GRP_I = 8;
__device__ __forceinline__ void grouped_wgrad_tile(const int n, const int c, const int g, const int gw, const int chunk,
                                                   const float *__restrict__ gY, const float *__restrict__ X,
                                                   const float *__restrict__ sw, float *__restrict__ part, const int rec,
                                                   const int bx, const int bg) {
    extern __shared__ float4 grp_lds4[];
    const int q = c >> 2, units = gw * q, R = max(1, TPB / units);
    const int tid = threadIdx.x, rs = tid / units, u = tid - rs * units;
    const bool active = rs < R;
    const int gl = u / q, qi = u - gl * q, grp = bg * gw + gl;
    const bool live = active && grp < g;
    float4 acc[GRP_I];
    float bacc[GRP_I];
#pragma unroll
    for (int i = 0; i < GRP_I; ++i) { acc[i] = make_float4(0.f, 0.f, 0.f, 0.f); bacc[i] = 0.f; }
    const long long r0 = (long long)bx * chunk, r1 = (r0 + chunk) < (long long)n ? (r0 + chunk) : (long long)n;
    if (live) {
        const float *xa = X + (size_t)grp * c + 4 * qi;  // + r * g * c
        const float *ya = gY + (size_t)grp * GRP_I;       // + r * c
        const float *sa = sw + grp;                       // + r * g
        const size_t xs = (size_t)g * c;
        constexpr int U = 4;  // rows in flight per thread
        long long r = r0 + rs;
        for (; r + (long long)(U - 1) * R < r1; r += (long long)U * R) {
            float4 a[U], y0[U], y1[U];
            float s[U];
#pragma unroll
            for (int t = 0; t < U; ++t) {
                const size_t rr = (size_t)(r + (long long)t * R);
                a[t] = *(const float4 *)(xa + rr * xs);
                y0[t] = *(const float4 *)(ya + rr * c);
                y1[t] = *(const float4 *)(ya + rr * c + 4);
                s[t] = qi == 0 ? sa[rr * g] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < U; ++t) {
                const float y[GRP_I] = {y0[t].x, y0[t].y, y0[t].z, y0[t].w, y1[t].x, y1[t].y, y1[t].z, y1[t].w};
#pragma unroll
                for (int i = 0; i < GRP_I; ++i) {
                    acc[i].x = __builtin_fmaf(y[i], a[t].x, acc[i].x); acc[i].y = __builtin_fmaf(y[i], a[t].y, acc[i].y);
                    acc[i].z = __builtin_fmaf(y[i], a[t].z, acc[i].z); acc[i].w = __builtin_fmaf(y[i], a[t].w, acc[i].w);
                    bacc[i] = __builtin_fmaf(y[i], s[t], bacc[i]);
                }
            }
        }
        for (; r < r1; r += R) {
            const size_t rr = (size_t)r;
            const float4 a = *(const float4 *)(xa + rr * xs), y0 = *(const float4 *)(ya + rr * c), y1 = *(const float4 *)(ya + rr * c + 4);
            const float sv = qi == 0 ? sa[rr * g] : 0.f;
            const float y[GRP_I] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
            for (int i = 0; i < GRP_I; ++i) {
                acc[i].x = __builtin_fmaf(y[i], a.x, acc[i].x); acc[i].y = __builtin_fmaf(y[i], a.y, acc[i].y);
                acc[i].z = __builtin_fmaf(y[i], a.z, acc[i].z); acc[i].w = __builtin_fmaf(y[i], a.w, acc[i].w);
                bacc[i] = __builtin_fmaf(y[i], sv, bacc[i]);
            }
        }
    }
    // row slots 1 .. R-1 through LDS, added to slot 0 in slot order: [slot][i][unit] float4, then [slot][i][gl] floats
    float4 *sAcc = grp_lds4;                                   // [R][GRP_I][units]
    float *sB = (float *)(sAcc + (size_t)R * GRP_I * units);   // [R][GRP_I][gw]
    if (active) {
#pragma unroll
        for (int i = 0; i < GRP_I; ++i) {
            sAcc[((size_t)rs * GRP_I + i) * units + u] = acc[i];
            if (qi == 0) sB[((size_t)rs * GRP_I + i) * gw + gl] = bacc[i];
        }
    }
    __syncthreads();
    if (rs == 0 && grp < g) {
        float *p = part + (size_t)bx * rec;
#pragma unroll
        for (int i = 0; i < GRP_I; ++i) {
            float4 v = sAcc[(size_t)i * units + u];
            for (int t = 1; t < R; ++t) {
                const float4 w = sAcc[((size_t)t * GRP_I + i) * units + u];
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            *(float4 *)(p + ((size_t)grp * GRP_I + i) * c + 4 * qi) = v;
            if (qi == 0) {
                float b = sB[(size_t)i * gw + gl];
                for (int t = 1; t < R; ++t) b += sB[((size_t)t * GRP_I + i) * gw + gl];
                p[(size_t)g * GRP_I * c + (size_t)grp * GRP_I + i] = b;
            }
        }
    }
}
inline size_t grouped_lds_bytes(int c, int gw) {
    const int units = gw * (c >> 2), R = std::max(1, TPB / units);
    return sizeof(float4) * (size_t)R * GRP_I * units + sizeof(float) * (size_t)R * GRP_I * gw;
}
// one call: the job by value
__global__ __launch_bounds__(TPB) void grouped_wgrad_kernel(WgradJob J) {
    const int bx = (int)blockIdx.x % J.chunks, bg = (int)blockIdx.x / J.chunks;
    grouped_wgrad_tile(J.n, J.cin, J.batch, J.gw, J.chunk, J.gY, J.X, J.rowscale, J.part, J.rec, bx, bg);
}
__global__ __launch_bounds__(TPB) void grouped_wgrad_kernel_jobs(const WgradJob *__restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].wg0) ++j;
    const WgradJob &J = jobs[j];
    const int local = (int)blockIdx.x - J.wg0;
    grouped_wgrad_tile(J.n, J.cin, J.batch, J.gw, J.chunk, J.gY, J.X, J.rowscale, J.part, J.rec, local % J.chunks, local / J.chunks);
}

template <bool BF16>
__global__ __launch_bounds__(TPB) void linear_wgrad_kernel_jobs(const WgradJob *__restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].wg0) ++j;
    const WgradJob &J = jobs[j];
    const int local = (int)blockIdx.x - J.wg0;
    const int bx = local % J.chunks, rest = local / J.chunks, by = rest % J.tiles, bz = rest / J.tiles;
    const bool multi = J.count > 0;
    wgrad_direct_tile<BF16>(J.n, J.cout, J.cin, J.tiles_i, multi ? J.mgY[bz] : J.gY + (long long)bz * J.sy, J.ldy,
                            multi ? J.mX[bz] : J.X + (long long)bz * J.sx, J.ldx, J.part, J.has_pb != 0, J.batch,
                            multi ? J.mxsc[bz] : nullptr, multi ? J.mxsh[bz] : nullptr, J.chunk, bx, by, bz);
}

// the records of all jobs -> their outputs.  A job's slots are whole workgroups.  With up to 32 chunk records a thread sums one
// output element; with more (the full-resolution jobs have hundreds of records for a few thousand outputs: one thread per output
// walked them as a chain of dependent loads, 75 us per launch) the four wavefronts of a workgroup take every fourth record of the
// same 64 consecutive elements -- every load instruction reads 256 contiguous bytes of one record (16 lanes per element, each on
// a record of its own, touched 16 lines per instruction for 16 bytes of each: 43 us per launch on average) -- and their sums
// meet in LDS, added in wavefront order.  Double accumulation; written where the job's own finalize would have written it.
__global__ __launch_bounds__(256) void wgrad_jobs_finalize_kernel(const WgradJob *__restrict__ jobs, int njobs, int total) {
    __shared__ double s_sum[3][64];
    const int sidx = blockIdx.x * 256 + threadIdx.x;  // (total is a multiple of 256, and so is every job's fin0)
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x * 256 >= jobs[j + 1].fin0) ++j;  // (uniform)
    const WgradJob &J = jobs[j];
    const int lanes = J.fin_lanes, local = sidx - J.fin0;
    const int rg = lanes == 1 ? 0 : (int)threadIdx.x >> 6;
    const int col = lanes == 1 ? local : (local >> 8) * 64 + ((int)threadIdx.x & 63);
    const int nblk = J.chunks;
    const size_t len = (size_t)J.rec;
    const float *part = J.part;
    const bool ok = col < J.rec;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (ok) {
        int b = rg;
        for (; b + 3 * lanes < nblk; b += 4 * lanes) {
            a0 += (double)part[(size_t)b * len + col];
            a1 += (double)part[(size_t)(b + lanes) * len + col];
            a2 += (double)part[(size_t)(b + 2 * lanes) * len + col];
            a3 += (double)part[(size_t)(b + 3 * lanes) * len + col];
        }
        for (; b < nblk; b += lanes) a0 += (double)part[(size_t)b * len + col];
    }
    double acc = (a0 + a1) + (a2 + a3);
    if (lanes != 1) {  // (uniform over the workgroup)
        if (rg > 0) s_sum[rg - 1][threadIdx.x & 63] = acc;
        __syncthreads();
        if (rg > 0) return;
        acc = (acc + s_sum[0][threadIdx.x]) + (s_sum[1][threadIdx.x] + s_sum[2][threadIdx.x]);
    }
    if (!ok) return;
    const float v = (float)acc;
    const int wlen = J.cout * J.cin, wtot = J.batch * wlen;
    if (J.count > 0) {  // MapWgradMulti
        if (col < wtot) {
            const int p = col / wlen;
            J.mdW[p][col - p * wlen] = v;
        } else {
            const int r = col - wtot, p = r / J.cout;
            if (J.mdb[p]) J.mdb[p][r - p * J.cout] = v;
        }
    } else if (col < wtot) {
        J.dW[col] = v;
    } else if (J.db) {
        J.db[col - wtot] = v;
    }
}

// operands 16-byte aligned with row strides that keep them so: the LDS-staged form applies (fp32 products only)
static bool wgrad_lds_ok(const void *a, long long ldy, long long sy, const void *b, long long ldx, long long sx) {
    return ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && ldy % 4 == 0 && ldx % 4 == 0 && sy % 4 == 0 && sx % 4 == 0;
}
static bool wgrad_lds_shape_ok(int cout, int cin) { return cout % 4 == 0 && cin % 4 == 0; }
constexpr size_t WL_LDS_BYTES = sizeof(float) * std::max<size_t>(4 * (size_t)WL_ROWS * WG_TILE,
                                                                 (size_t)(TPB / WAVE) * (WG_MT * WG_MT * 4 + WG_MT) * (WAVE + 1));

// ----------------------------------------------------------- skinny projection --
// y[n,o] = sum_i x[n,i] W[o,i] for cout <= 64 (the G-wide projections kW, qW of the attention logits); the
// BLAS kernel chosen for an N x 48 x 6 product runs 190 us (profiles/r01_fused_v5_*).  One lane per output,
// W in LDS, the x row is shared by the cout lanes of a point.
// xsc / xsh != NULL: the input row passes through ReLU(x * xsc + xsh) first (BatchNorm + ReLU of linear_q / linear_k
// fused into the projection that consumes them)
// blockIdx.y == 1 works on the second operand set (x2, xsc2, xsh2 -> y2; same W): the key and query projections
__global__ __launch_bounds__(TPB) void skinny_fwd_kernel(long long n, int cin, int cout, const float *x, const float *__restrict__ W,
                                                         const float *xsc, const float *xsh, float *y, const float *x2,
                                                         const float *xsc2, const float *xsh2, float *y2) {
    extern __shared__ float4 lds4[];
    if (blockIdx.y) { x = x2; xsc = xsc2; xsh = xsh2; y = y2; }
    float *sW = (float *)lds4;  // [cout][cin + 4], then [2][cin] scale / shift
    const int ldw = cin + 4, cq = cin >> 2;
    float *sSc = sW + (size_t)cout * ldw, *sSh = sSc + cin;
    for (int e = threadIdx.x; e < cout * cq; e += TPB) {
        const int r = e / cq, q = e - r * cq;
        *(float4 *)(sW + (size_t)r * ldw + 4 * q) = *(const float4 *)(W + (size_t)r * cin + 4 * q);
    }
    if (xsc)
        for (int e = threadIdx.x; e < cin; e += TPB) { sSc[e] = xsc[e]; sSh[e] = xsh[e]; }
    __syncthreads();
    const long long total = n * cout;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const long long row = e / cout;
        const int o = (int)(e - row * cout);
        const float4 *xr = (const float4 *)(x + row * cin), *wr = (const float4 *)(sW + (size_t)o * ldw);
        float acc = 0.f;
        if (xsc) {
            int q = 0;
            for (; q + 4 <= cq; q += 4) {  // four row quads in flight per trip (cq is a multiple of 4 for C = 48 ... 512)
                float4 av[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) av[u] = xr[q + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 a = av[u];
                    const float4 w = wr[q + u], s4 = ((const float4 *)sSc)[q + u], h4 = ((const float4 *)sSh)[q + u];
                    a.x = fmaxf(__builtin_fmaf(a.x, s4.x, h4.x), 0.f); a.y = fmaxf(__builtin_fmaf(a.y, s4.y, h4.y), 0.f);
                    a.z = fmaxf(__builtin_fmaf(a.z, s4.z, h4.z), 0.f); a.w = fmaxf(__builtin_fmaf(a.w, s4.w, h4.w), 0.f);
                    acc = __builtin_fmaf(a.x, w.x, acc); acc = __builtin_fmaf(a.y, w.y, acc);
                    acc = __builtin_fmaf(a.z, w.z, acc); acc = __builtin_fmaf(a.w, w.w, acc);
                }
            }
            for (; q < cq; ++q) {
                float4 a = xr[q];
                const float4 w = wr[q], s4 = ((const float4 *)sSc)[q], h4 = ((const float4 *)sSh)[q];
                a.x = fmaxf(__builtin_fmaf(a.x, s4.x, h4.x), 0.f); a.y = fmaxf(__builtin_fmaf(a.y, s4.y, h4.y), 0.f);
                a.z = fmaxf(__builtin_fmaf(a.z, s4.z, h4.z), 0.f); a.w = fmaxf(__builtin_fmaf(a.w, s4.w, h4.w), 0.f);
                acc = __builtin_fmaf(a.x, w.x, acc); acc = __builtin_fmaf(a.y, w.y, acc);
                acc = __builtin_fmaf(a.z, w.z, acc); acc = __builtin_fmaf(a.w, w.w, acc);
            }
        } else {
            for (int q = 0; q < cq; ++q) {
                const float4 a = xr[q], w = wr[q];
                acc = __builtin_fmaf(a.x, w.x, acc); acc = __builtin_fmaf(a.y, w.y, acc);
                acc = __builtin_fmaf(a.z, w.z, acc); acc = __builtin_fmaf(a.w, w.w, acc);
            }
        }
        y[e] = acc;
    }
}

// gx[n,i] = sum_o gy[n,o] W[o,i]; one lane per float4 of gx
__global__ __launch_bounds__(TPB) void skinny_bwd_kernel(long long n, int cin, int cout, const float *gy,
                                                         const float *__restrict__ W, float *gx, const float *gy2, float *gx2,
                                                         int main_blocks, gva::PtvRiders Rs) {
    if ((int)blockIdx.x >= main_blocks) {  // trailing workgroups: deferred parameter-gradient sums (gva_common.h, riders)
        if (blockIdx.y == 0) gva::rider_run(Rs, (int)blockIdx.x - main_blocks);
        return;
    }
    if (blockIdx.y) { gy = gy2; gx = gx2; }
    const int cq = cin >> 2;
    const long long total = n * cq;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)main_blocks * TPB) {
        const long long row = e / cq;
        const int q = (int)(e - row * cq);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // six outputs per trip (G = 6, 12, 24, 48 ...): their twelve loads are in flight together -- one output per trip
        // waited for its own two loads every time, and the kernel sat parked on memory for 89 % of its wave cycles
        // (profiles/r02_final_sq_counters.jsonl)
        const float *g = gy + row * cout;
        const float *wc = W + 4 * q;
        int o = 0;
        for (; o + 6 <= cout; o += 6) {
            float sv[6];
            float4 wv[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) { sv[u] = g[o + u]; wv[u] = *(const float4 *)(wc + (size_t)(o + u) * cin); }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                acc.x = __builtin_fmaf(sv[u], wv[u].x, acc.x); acc.y = __builtin_fmaf(sv[u], wv[u].y, acc.y);
                acc.z = __builtin_fmaf(sv[u], wv[u].z, acc.z); acc.w = __builtin_fmaf(sv[u], wv[u].w, acc.w);
            }
        }
        for (; o < cout; ++o) {
            const float s = g[o];
            const float4 w = *(const float4 *)(wc + (size_t)o * cin);
            acc.x = __builtin_fmaf(s, w.x, acc.x); acc.y = __builtin_fmaf(s, w.y, acc.y);
            acc.z = __builtin_fmaf(s, w.z, acc.z); acc.w = __builtin_fmaf(s, w.w, acc.w);
        }
        ((float4 *)gx)[e] = acc;
    }
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace dense

using namespace dense;

extern "C" size_t dense_workspace_bytes(int n, int cout, int cin) {  // cout*cin = total outputs over all batches
    const size_t chunks = (size_t)(n + WG_CHUNK_MIN - 1) / WG_CHUNK_MIN + 1;
    const size_t wg = sizeof(float) * chunks * ((size_t)cout * cin + cout);
    const size_t bn = sizeof(float) * (size_t)MAX_BLK * 2 * (size_t)std::max(cout, cin);
    return align_up(std::max(wg, bn)) + 1024;
}

// rows per split-K workgroup of the weight gradient
// `tiles` = output tiles x products of the launch.  A workgroup costs ~10 us of fixed work (operand latency, the
// 40 KB cross-wave reduction, its partial record) however few rows it sums, and three fit a compute unit: at the deep
// levels (n ~ 4 500, 80 tile-products) 256-row chunks made 1 440 workgroups = two full rounds of that fixed cost for 18
// records to finalize.  Target ~3 workgroups per compute unit in ONE round; never fewer than 256 rows per workgroup.
static int wg_chunk(int n, int tiles, bool filed = false) {
    // filed: the launch will run inside the batched launch of a whole backward (WgradJob), where the OTHER jobs fill the GPU:
    // longer chunks -- fewer records to write and to sum, the per-workgroup fixed cost paid less often
    // (bench step at 768 / 384 / 192 / 96 workgroups per job: 10.51 / 10.42 / 10.42 / 10.47 ms)
    const int target = filed ? 256 : 768;  // (768: swept in round 2, DESIGN.md / profiles/HISTORY.md)
    const int chunks = std::max(1, target / std::max(1, tiles));
    const long long rows = ((long long)n + chunks - 1) / chunks;
    long long chunk = std::max<long long>(WG_CHUNK, (rows + 127) / 128 * 128);
    // The grid is (chunks, tiles, products), x fastest, and workgroups go round-robin over the 8 XCDs: with a chunk count that
    // is a multiple of 8 all tile / product workgroups of a chunk -- which read the same operand rows at the same time --
    // land on ONE XCD and queue on the same L2 lines (measured with a grid built that way on purpose: 11.34 against 10.98 ms
    // per step).  Keep the count off the multiples of 8 when more than one workgroup shares a chunk.
    // (forcing the count ODD -- consecutive tiles walking through all eight XCDs -- measured 0.07-0.09 ms slower than the
    // counts the rule above produces: 469 / 235 / 134, 74 / 50 / 37, 18 / 12 / 9, 5 / 3 / 2 at the four levels)
    if (tiles > 1) {
        int guard = 0;
        while ((((long long)n + chunk - 1) / chunk) % 8 == 0 && ((long long)n + chunk - 1) / chunk > 1 && guard++ < 16) chunk += 128;
    }
    return (int)std::min<long long>(chunk, 1 << 20);
}

static int bn_grid(int n, int c) {
    const int rl = std::max(1, TPB / (c >> 2));
    long long b = ((long long)n + rl * 4 - 1) / (rl * 4);  // (8 rows per lane: the same; 16: +0.1 ms per step)
    // deep levels: at most 128 records, which the apply kernel's workgroups then sum themselves (bn_bwd_finapply_kernel)
    return (int)std::max<long long>(1, std::min<long long>(b, n <= 16384 ? 128 : MAX_BLK));
}

// statistics pass + finalize; gamma / beta / sc / sh != NULL additionally emit the folded affine
static int bn_stats_impl(int n, int c, const float *x, float *mean, float *rstd, float *running_mean, float *running_var,
                         long long *num_batches_tracked, float eps, float momentum, const float *gamma, const float *beta,
                         float *sc, float *sh, void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 1 || c < 4 || c % 4 != 0 || c > 1024) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < dense_workspace_bytes(n, c, c)) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = bn_grid(n, c);
    float *part = (float *)workspace;
    {
        PtvScopedTimer t(KID_BN_STATS, st, 4.0 * n * c);
        hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(TPB), sizeof(float4) * 2 * TPB, st, n, c, x, part);
    }
    if (nblk >= 64)
        hipLaunchKernelGGL(bn_finalize_kernel<16>, dim3((c + 15) / 16), dim3(1024), 0, st, (const float *)part, nblk, c, n, x, eps,
                           momentum, mean, rstd, running_mean, running_var, num_batches_tracked, gamma, beta, sc, sh);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<64>, dim3((c + 63) / 64), dim3(1024), 0, st, (const float *)part, nblk, c, n, x, eps,
                           momentum, mean, rstd, running_mean, running_var, num_batches_tracked, gamma, beta, sc, sh);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int bn_stats_hip_launcher(int n, int c, const float *x, float *mean, float *rstd, float *running_mean,
                                     float *running_var, long long *num_batches_tracked, float eps, float momentum,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    return bn_stats_impl(n, c, x, mean, rstd, running_mean, running_var, num_batches_tracked, eps, momentum, nullptr, nullptr,
                         nullptr, nullptr, workspace, workspace_bytes, stream);
}

// bn_stats that also emits the folded affine (sc = rstd * gamma, sh = beta - mean * sc) for consumers that apply the
// normalisation on their operand load (rows_gemm_fused, linear_wgrad_multi, skinny_linear_forward)
extern "C" int bn_stats_affine_hip_launcher(int n, int c, const float *x, const float *gamma, const float *beta, float *mean,
                                            float *rstd, float *sc, float *sh, float *running_mean, float *running_var,
                                            long long *num_batches_tracked, float eps, float momentum, void *workspace,
                                            size_t workspace_bytes, void *stream) {
    if (!gamma || !beta || !sc || !sh) return PTV2_ERR_ARG;
    return bn_stats_impl(n, c, x, mean, rstd, running_mean, running_var, num_batches_tracked, eps, momentum, gamma, beta, sc, sh,
                         workspace, workspace_bytes, stream);
}

// first level for many records: block (x, y) folds records y, y + gridDim.y, ... of 64 columns into ONE record of
// the same form (sum; centred sum of squares; its row count is implied by the records it covers)
__global__ __launch_bounds__(gva::FIN_COLS *gva::FIN_SLICES) void bn_fold_tiles_kernel(BnTileSet A, BnTileSet B, int nrb, int c,
                                                                                       int n) {
    __shared__ double s1[gva::FIN_SLICES][gva::FIN_COLS], s2[gva::FIN_SLICES][gva::FIN_COLS];
    const BnTileSet &S = blockIdx.z ? B : A;
    const float *__restrict__ part = S.part;
    double *__restrict__ out = S.fold;
    const int col = threadIdx.x & (gva::FIN_COLS - 1), sl = threadIdx.x / gva::FIN_COLS;
    const int ch = blockIdx.x * gva::FIN_COLS + col;
    double a = 0.0, b = 0.0;
    if (ch < c) {
        for (int k = blockIdx.y * gva::FIN_SLICES + sl; k < nrb; k += gridDim.y * gva::FIN_SLICES) {
            const int cnt = bn_tile_rows(S, k, n);
            const double sb = (double)part[(size_t)k * 2 * c + ch];
            a += sb;
            b += (double)part[(size_t)k * 2 * c + c + ch] + sb * sb / (double)cnt;
        }
    }
    s1[sl][col] = a;
    s2[sl][col] = b;
    __syncthreads();
    if (sl == 0 && ch < c) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int t = 0; t < gva::FIN_SLICES; ++t) { t1 += s1[t][col]; t2 += s2[t][col]; }
        out[(size_t)blockIdx.y * 2 * c + ch] = t1;       // sum
        out[(size_t)blockIdx.y * 2 * c + c + ch] = t2;   // sum_b (M2_b + S_b^2 / n_b): only "- n mean^2" is missing
    }
}

// second level: nrec folded records (float64) -> mean, rstd, folded affine, running buffers
__global__ void bn_finalize_folded_kernel(BnTileSet A, BnTileSet B, int nrec, int c, int n, float eps, float momentum) {
    const BnTileSet &S = blockIdx.z ? B : A;
    const double *__restrict__ rec = S.fold;
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double t1 = 0.0, t2 = 0.0;
    for (int k = 0; k < nrec; ++k) { t1 += rec[(size_t)k * 2 * c + ch]; t2 += rec[(size_t)k * 2 * c + c + ch]; }
    bn_tiles_emit(S, ch, t1, t2, n, eps, momentum);
}

// statistics of a (n,c) tensor from the records its producing rows_gemm_fused launch left in `part`
extern "C" size_t bn_tiles_floats(int n, int c) {  // floats of a statistics record buffer (incl. the folding scratch)
    return (size_t)((n + 63) / 64) * 2 * c + 2 + 2 * (size_t)16 * 2 * c;
}
size_t bn_tiles_floats_rb(int n, int c, int rb) {  // the same for records of rb rows
    return (size_t)((n + rb - 1) / rb) * 2 * c + 2 + 2 * (size_t)16 * 2 * c;
}

// count (1 or 2) tensors of one shape in one launch (two for > 512 records: fold, then finish)
static int bn_tiles_finalize_sets(int n, int c, int count, BnTileSet *sets, float eps, float momentum, void *stream, int rb = 64) {
    const int nrb_all = (n + rb - 1) / rb;
    for (int i = 0; i < count; ++i) {
        BnTileSet &S = sets[i];
        S.rb = rb;
        if (!S.part || !S.mean || !S.rstd || ((S.sc != nullptr) && (!S.gamma || !S.beta || !S.sh))) return PTV2_ERR_ARG;
        // the folded records live behind the tile records (the GEMM wrote nrb * 2c floats; 16 * 2c doubles more are reserved)
        S.fold = (double *)(const_cast<float *>(S.part) + (((size_t)nrb_all * 2 * c + 1) & ~(size_t)1));
    }
    const BnTileSet A = sets[0], B = sets[count - 1];
    const unsigned cb = (unsigned)((c + gva::FIN_COLS - 1) / gva::FIN_COLS);
    if (nrb_all > 4096) {  // two levels: 16 folding blocks per 64 columns, then a one-thread-per-column finish
        const int ny = 16;
        hipLaunchKernelGGL(bn_fold_tiles_kernel, dim3(cb, ny, count), dim3(gva::FIN_COLS * gva::FIN_SLICES), 0, (hipStream_t)stream,
                           A, B, nrb_all, c, n);
        hipLaunchKernelGGL(bn_finalize_folded_kernel, dim3((c + 63) / 64, 1, count), dim3(64), 0, (hipStream_t)stream, A, B, ny, c, n,
                           eps, momentum);
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    if (nrb_all >= 1024 && ((c + 15) / 16) * count <= 32) {
        unsigned *cnt = ptv2_stream_counters((hipStream_t)stream);
        if (!cnt) return PTV2_ERR_LAUNCH;
        hipLaunchKernelGGL(bn_finalize_tiles_split_kernel, dim3((c + 15) / 16, BNT_NS, count), dim3(1024), 0, (hipStream_t)stream, A, B,
                           nrb_all, c, n, eps, momentum, cnt + CNT_BN_TILES);
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    if (nrb_all >= 64)  // many records: 16 columns x 64 record slices per workgroup
        hipLaunchKernelGGL(bn_finalize_tiles_kernel<16>, dim3((c + 15) / 16, 1, count), dim3(1024), 0, (hipStream_t)stream, A, B,
                           nrb_all, c, n, eps, momentum);
    else
        hipLaunchKernelGGL(bn_finalize_tiles_kernel<64>, dim3(cb, 1, count), dim3(1024), 0, (hipStream_t)stream, A, B, nrb_all, c, n,
                           eps, momentum);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int bn_tiles_finalize_hip_launcher(int n, int c, float *part, const float *gamma, const float *beta,
                                              float *mean, float *rstd, float *sc, float *sh, float *running_mean,
                                              float *running_var, long long *num_batches_tracked, float eps, float momentum,
                                              void *stream) {
    if (n < 1 || c < 4) return PTV2_ERR_ARG;
    BnTileSet S{part, mean, rstd, running_mean, running_var, num_batches_tracked, gamma, beta, sc, sh, nullptr, 64};
    return bn_tiles_finalize_sets(n, c, 1, &S, eps, momentum, stream);
}
// internal (block.hip): records of rb rows each (bn_tiles_floats_rb floats) -- the attention's tile kernel leaves 16-row records
int bn_tiles_finalize_rb(int n, int c, int rb, float *part, const float *gamma, const float *beta, float *mean, float *rstd, float *sc,
                         float *sh, float *running_mean, float *running_var, long long *num_batches_tracked, float eps, float momentum,
                         void *stream) {
    if (n < 1 || c < 4 || (rb != 16 && rb != 64)) return PTV2_ERR_ARG;
    BnTileSet S{part, mean, rstd, running_mean, running_var, num_batches_tracked, gamma, beta, sc, sh, nullptr, rb};
    return bn_tiles_finalize_sets(n, c, 1, &S, eps, momentum, stream, rb);
}

// two tensors of one shape (internal to the block runtime: the q / k BatchNorms); arrays of 2
int bn_tiles_finalize_pair(int n, int c, float *const *part, const float *const *gamma, const float *const *beta,
                           float *const *mean, float *const *rstd, float *const *sc, float *const *sh, float *const *running_mean,
                           float *const *running_var, long long *const *num_batches_tracked, float eps, float momentum,
                           void *stream, int rb) {
    if (n < 1 || c < 4 || (rb != 16 && rb != 64)) return PTV2_ERR_ARG;
    BnTileSet S[2];
    for (int i = 0; i < 2; ++i)
        S[i] = BnTileSet{part[i], mean[i], rstd[i], running_mean[i], running_var[i], num_batches_tracked[i], gamma[i], beta[i], sc[i],
                         sh[i], nullptr, rb};
    return bn_tiles_finalize_sets(n, c, 2, S, eps, momentum, stream, rb);
}

// internal (block.hip): BatchNorm statistics from the producing GEMM's tile records AND the Block tail
// y = ReLU(residual + rowscale * BN(x)) in one launch when the records are few (deep levels); returns 0 when it declines
int bn_tiles_apply_residual(int n, int c, float *part, const float *gamma, const float *beta, float *mean, float *rstd, float *sc,
                            float *sh, float *running_mean, float *running_var, long long *num_batches_tracked, float eps,
                            float momentum, const float *x, const float *residual, const float *rowscale, float *y, void *stream,
                            int rb) {
    const int nrb = (n + rb - 1) / rb;
    const char *e = getenv("AO_AMD_BN_FINAPPLY");
    if (nrb > (rb == 16 ? 512 : 256) || c % 4 != 0 || (e && e[0] == '0') || (rb != 16 && rb != 64)) return 0;
    BnTileSet S{part, mean, rstd, running_mean, running_var, num_batches_tracked, gamma, beta, sc, sh, nullptr, rb};
    const dim3 grid((unsigned)((c + FA_COLS - 1) / FA_COLS), (unsigned)((n + FA_ROWS - 1) / FA_ROWS));
    {
        PtvScopedTimer t(KID_BN_APPLY, (hipStream_t)stream, 12.0 * n * c);
        hipLaunchKernelGGL(bn_tiles_apply_residual_kernel<0>, grid, dim3(TPB), 0, (hipStream_t)stream, S, nrb, n, c, eps, momentum, x,
                           residual, rowscale, y);
    }
    return 1;
}

// internal (model.hip): the same for y = ReLU(BN(x)) -- statistics from the producing GEMM's 64-row records and the apply pass in
// one launch (was bn_stats + bn_finalize + bn_apply); returns 0 when it declines (many records: the three launches stay)
int bn_tiles_apply_relu(int n, int c, float *part, const float *gamma, const float *beta, float *mean, float *rstd, float *running_mean,
                        float *running_var, long long *num_batches_tracked, float eps, float momentum, const float *x, float *y,
                        void *stream) {
    const int nrb = (n + 63) / 64;
    const char *e = getenv("AO_AMD_BN_FINAPPLY");
    if (nrb > 512 || c % 4 != 0 || (e && e[0] == '0')) return 0;
    BnTileSet S{part, mean, rstd, running_mean, running_var, num_batches_tracked, gamma, beta, nullptr, nullptr, nullptr, 64};
    const dim3 grid((unsigned)((c + FA_COLS - 1) / FA_COLS), (unsigned)((n + FA_ROWS - 1) / FA_ROWS));
    {
        PtvScopedTimer t(KID_BN_APPLY, (hipStream_t)stream, 8.0 * n * c);
        hipLaunchKernelGGL(bn_tiles_apply_residual_kernel<1>, grid, dim3(TPB), 0, (hipStream_t)stream, S, nrb, n, c, eps, momentum, x,
                           (const float *)nullptr, (const float *)nullptr, y);
    }
    return 1;
}

extern "C" int bn_apply_hip_launcher(int n, int c, const float *x, const float *mean, const float *rstd,
                                     const float *gamma, const float *beta, int relu, float *y, void *stream) {
    if (n < 0 || c < 4 || c % 4 != 0) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const long long total4 = (long long)n * (c >> 2);
    const int nblk = (int)std::min<long long>((total4 + TPB - 1) / TPB, 256 * 16);
    {
        PtvScopedTimer t(KID_BN_APPLY, (hipStream_t)stream, 8.0 * n * c);
        hipLaunchKernelGGL(bn_apply_kernel, dim3(nblk), dim3(TPB), 0, (hipStream_t)stream, total4, c >> 2, x, mean, rstd,
                           gamma, beta, relu, y);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// training-mode forward as one call (statistics + running buffers, then the apply pass; residual != NULL selects
// the Block tail y = ReLU(residual + rowscale * BN(x)))
extern "C" int bn_forward_hip_launcher(int n, int c, const float *x, const float *gamma, const float *beta, int relu,
                                       float *mean, float *rstd, float *running_mean, float *running_var,
                                       long long *num_batches_tracked, float eps, float momentum, const float *residual,
                                       const float *rowscale, float *y, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    const int rc = bn_stats_hip_launcher(n, c, x, mean, rstd, running_mean, running_var, num_batches_tracked, eps, momentum,
                                         workspace, workspace_bytes, stream);
    if (rc != PTV2_OK) return rc;
    if (residual) return bn_apply_residual_hip_launcher(n, c, x, mean, rstd, gamma, beta, residual, rowscale, y, stream);
    return bn_apply_hip_launcher(n, c, x, mean, rstd, gamma, beta, relu, y, stream);
}

extern "C" int bn_apply_residual_hip_launcher(int n, int c, const float *x, const float *mean, const float *rstd,
                                              const float *gamma, const float *beta, const float *residual,
                                              const float *rowscale, float *y, void *stream) {
    if (n < 0 || c < 4 || c % 4 != 0 || !residual) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const long long total4 = (long long)n * (c >> 2);
    const int nblk = (int)std::min<long long>((total4 + TPB - 1) / TPB, 256 * 16);
    {
        PtvScopedTimer t(KID_BN_APPLY, (hipStream_t)stream, 12.0 * n * c);
        hipLaunchKernelGGL(bn_apply_residual_kernel, dim3(nblk), dim3(TPB), 0, (hipStream_t)stream, total4, c >> 2, x, mean,
                           rstd, gamma, beta, residual, rowscale, y);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int bn_backward_residual_hip_launcher(int n, int c, const float *x, const float *gy, const float *y,
                                                 const float *rowscale, const float *mean, const float *rstd,
                                                 const float *gamma, int training, float *gx, float *g_residual,
                                                 float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                                                 void *stream) {
    if (n < 1 || c < 4 || c % 4 != 0 || c > 1024) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < dense_workspace_bytes(n, c, c)) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = bn_grid(n, c);
    float *part = (float *)workspace;
    {
        PtvScopedTimer t(KID_BN_BWD_REDUCE, st, 12.0 * n * c);
        hipLaunchKernelGGL(bn_bwd_reduce_residual_kernel, dim3(nblk), dim3(TPB), sizeof(float4) * 2 * TPB, st, n, c, x, gy, y,
                           rowscale, mean, rstd, part);
    }
    if (finapply_ok(n, nblk)) {
        PtvScopedTimer t(KID_BN_BWD_FINAPPLY, st, 20.0 * n * c);
        const BnFinApply A{part, nblk, 2 * c, 0, x, gy, mean, rstd, gamma, nullptr, gx, dbeta, dgamma, y, rowscale, g_residual};
        launch_finapply(st, n, c, 1, training, true, 1, A, A);
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    launch_finalize(st, (const float *)part, nblk, 2 * c, gva::MapSplit2<float>{dbeta, dgamma, c});
    const long long total4 = (long long)n * (c >> 2);
    const int nb2 = (int)std::min<long long>((total4 + TPB - 1) / TPB, 256 * 16);
    {
        PtvScopedTimer t(KID_BN_BWD_APPLY_RES, st, 20.0 * n * c);
        hipLaunchKernelGGL(bn_bwd_apply_residual_kernel, dim3(nb2), dim3(TPB), 0, st, total4, c >> 2, 1.0f / (float)n, x, gy, y,
                           rowscale, mean, rstd, gamma, (const float *)dbeta, (const float *)dgamma, training, gx, g_residual);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int bn_backward_hip_launcher(int n, int c, const float *x, const float *gy, const float *mean,
                                        const float *rstd, const float *gamma, const float *beta, int relu,
                                        int training, float *gx, float *dgamma, float *dbeta, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    if (n < 1 || c < 4 || c % 4 != 0 || c > 1024) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < dense_workspace_bytes(n, c, c)) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = bn_grid(n, c);
    float *part = (float *)workspace;
    {
        PtvScopedTimer t(KID_BN_BWD_REDUCE, st, 8.0 * n * c);
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(TPB), sizeof(float4) * 2 * TPB, st, n, c, x, gy, mean,
                           rstd, gamma, beta, relu, part, BnSecond{});
    }
    if (finapply_ok(n, nblk)) {
        PtvScopedTimer t(KID_BN_BWD_FINAPPLY, st, 12.0 * n * c);
        const BnFinApply A{part, nblk, 2 * c, 0, x, gy, mean, rstd, gamma, beta, gx, dbeta, dgamma, nullptr, nullptr, nullptr};
        launch_finapply(st, n, c, relu, training, false, 1, A, A);
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    launch_finalize(st, (const float *)part, nblk, 2 * c, gva::MapSplit2<float>{dbeta, dgamma, c});
    const long long total4 = (long long)n * (c >> 2);
    const int nb2 = (int)std::min<long long>((total4 + TPB - 1) / TPB, 256 * 16);
    {
        PtvScopedTimer t(KID_BN_BWD_APPLY, st, 12.0 * n * c);
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb2), dim3(TPB), 0, st, total4, c >> 2, 1.0f / (float)n, x, gy, mean,
                           rstd, gamma, beta, relu, (const float *)dbeta, (const float *)dgamma, training, gx, BnSecond{});
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// bn_backward whose reduce pass already ran in the epilogue of the GEMM that produced gy (rows_gemm_bnbwd_hip_launcher left
// nrec records of [2][c] in `records`): finalize + apply only
extern "C" int bn_backward_records_hip_launcher(int n, int c, const float *x, const float *gy, const float *mean,
                                                const float *rstd, const float *gamma, const float *beta, int relu,
                                                int training, float *gx, float *dgamma, float *dbeta, const float *records,
                                                int nrec, void *stream) {
    if (n < 1 || c < 4 || c % 4 != 0 || c > 1024 || !records || nrec < 1) return PTV2_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (finapply_ok(n, nrec)) {
        PtvScopedTimer t(KID_BN_BWD_FINAPPLY, st, 12.0 * n * c);
        const BnFinApply A{records, nrec, 2 * c, 0, x, gy, mean, rstd, gamma, beta, gx, dbeta, dgamma, nullptr, nullptr, nullptr};
        launch_finapply(st, n, c, relu, training, false, 1, A, A);
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    launch_finalize(st, records, nrec, 2 * c, gva::MapSplit2<float>{dbeta, dgamma, c});
    const long long total4 = (long long)n * (c >> 2);
    const int nb2 = (int)std::min<long long>((total4 + TPB - 1) / TPB, 256 * 16);
    {
        PtvScopedTimer t(KID_BN_BWD_APPLY, st, 12.0 * n * c);
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb2), dim3(TPB), 0, st, total4, c >> 2, 1.0f / (float)n, x, gy, mean,
                           rstd, gamma, beta, relu, (const float *)dbeta, (const float *)dgamma, training, gx, BnSecond{});
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

struct MapBnPair {  // record [dbeta0 c | dgamma0 c | dbeta1 c | dgamma1 c]
    float *db0, *dg0, *db1, *dg1;
    int c;
    __device__ void operator()(int j, double v) const {
        const int s = j / c, k = j - s * c;
        (s == 0 ? db0 : s == 1 ? dg0 : s == 2 ? db1 : dg1)[k] = (float)v;
    }
};

// ---- the q / k BatchNorms' reduce inside the launch that forms their output gradients (skinny_bn_bwd_reduce_kernel) ----
// block.hip arms this with the operands of the bn_backward_pair call that will follow its attention backward; gva_block.hip, at
// the skinny input-gradient launch, asks skinny_backward_pair_bn_reduce to run the fused kernel instead; the pair launcher finds
// `done` with matching operands and skips its reduce launch.  Thread-local: one Block backward per thread at a time.
struct SkinnyBnArm {
    bool armed = false, done = false;
    int n = 0, c = 0, nblk = 0, relu = 0;
    const float *x[2] = {}, *mean[2] = {}, *rstd[2] = {}, *gamma[2] = {}, *beta[2] = {};
    const float *gy[2] = {};
    float *part = nullptr;
};
static thread_local SkinnyBnArm t_skinny_bn;

void ptv2_skinny_bn_arm(int n, int c, const float *const *x, const float *const *gy, const float *const *mean, const float *const *rstd,
                        const float *const *gamma, const float *const *beta, int relu, void *workspace, size_t workspace_bytes) {
    SkinnyBnArm &K = t_skinny_bn;
    K = SkinnyBnArm{};
    static const bool off = [] { const char *e = getenv("AO_AMD_SKINNY_BN"); return e && e[0] == '0'; }();  // A/B switch
    if (off || n < 1 || c < 4 || c % 4 != 0 || (c >> 2) > TPB || !workspace || workspace_bytes < dense_workspace_bytes(n, 2 * c, c)) return;
    K.armed = true;
    K.n = n; K.c = c; K.nblk = bn_grid(n, c); K.relu = relu;
    // (deep levels: two rows per lane up to 256 records -- 128 records of seven rows per lane left the launch at one workgroup per
    // CU waiting on its own loads: 21.6 us at 4 501 rows x 192 against 15.3)
    constexpr int cap = 256;
    if (n <= 16384) {
        const int rl = std::max(1, TPB / (c >> 2));
        K.nblk = (int)std::max<long long>(1, std::min<long long>(((long long)n + rl * 2 - 1) / (rl * 2), cap));
    }
    for (int i = 0; i < 2; ++i) { K.x[i] = x[i]; K.gy[i] = gy[i]; K.mean[i] = mean[i]; K.rstd[i] = rstd[i]; K.gamma[i] = gamma[i]; K.beta[i] = beta[i]; }
    K.part = (float *)workspace;
}
void ptv2_skinny_bn_disarm(void) { t_skinny_bn.armed = false; }

int skinny_linear_backward_pair(int n, int cin, int cout, const float *const *gy, const float *W, float *const *gx, void *stream);
// internal (gva_block.hip): gx[i] = gy[i] W as skinny_linear_backward_pair -- and, when armed for exactly these outputs, the
// reduce records of the BatchNorm backward that consumes them, in the same launch
int skinny_backward_pair_bn_reduce(int n, int cin, int cout, const float *const *gy, const float *W, float *const *gx, void *stream) {
    SkinnyBnArm &K = t_skinny_bn;
    const bool fuse = K.armed && K.n == n && K.c == cin && K.gy[0] == gx[0] && K.gy[1] == gx[1] && n > 0;
    K.armed = false;
    if (!fuse) return skinny_linear_backward_pair(n, cin, cout, gy, W, gx, stream);
    hipStream_t st = (hipStream_t)stream;
    {
        PtvScopedTimer t(KID_SKINNY_BWD, st, 8.0 * n * (cin + cout) + 8.0 * n * cin);
        const gva::PtvRiders Rs = gva::ptv2_rider_take();
        const SkinnyBn A{gy[0], gx[0], K.x[0], K.mean[0], K.rstd[0], K.gamma[0], K.beta[0]};
        const SkinnyBn B{gy[1], gx[1], K.x[1], K.mean[1], K.rstd[1], K.gamma[1], K.beta[1]};
        hipLaunchKernelGGL(skinny_bn_bwd_reduce_kernel, dim3(K.nblk + gva::rider_blocks(Rs), 2), dim3(TPB), sizeof(float4) * 2 * TPB, st,
                           n, cin, cout, A, B, W, K.relu, K.part, K.nblk, Rs);
    }
    K.done = true;
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// two BatchNorm backwards of one shape (x[i], gy[i], ... i = 0, 1) in the three launches of one
// (workspace: dense_workspace_bytes(n, 2 * c, c))
extern "C" int bn_backward_pair_hip_launcher(int n, int c, const float *const *x, const float *const *gy,
                                             const float *const *mean, const float *const *rstd, const float *const *gamma,
                                             const float *const *beta, int relu, int training, float *const *gx,
                                             float *const *dgamma, float *const *dbeta, void *workspace, size_t workspace_bytes,
                                             void *stream) {
    if (n < 1 || c < 4 || c % 4 != 0 || c > 1024 || !x || !gy || !mean || !rstd || !gamma || !beta || !gx || !dgamma || !dbeta)
        return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < dense_workspace_bytes(n, 2 * c, c)) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int nblk = bn_grid(n, c);
    float *part = (float *)workspace;
    SkinnyBnArm &K = t_skinny_bn;
    const bool reduced = K.done && K.part == part && K.n == n && K.c == c && K.gy[0] == gy[0] && K.gy[1] == gy[1];
    K.done = false;
    if (reduced) nblk = K.nblk;
    const BnSecond sec{x[1], gy[1], mean[1], rstd[1], gamma[1], beta[1], gx[1], dgamma[1], dbeta[1]};
    if (!reduced) {  // (else: the records are there already, left by the launch that formed gy -- skinny_backward_pair_bn_reduce)
        PtvScopedTimer t(KID_BN_BWD_REDUCE, st, 16.0 * n * c);
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk, 2), dim3(TPB), sizeof(float4) * 2 * TPB, st, n, c, x[0], gy[0], mean[0],
                           rstd[0], gamma[0], beta[0], relu, part, sec);
    }
    if (finapply_ok(n, nblk)) {  // record of a block: [set 0: dbeta c | dgamma c][set 1: ...]
        PtvScopedTimer t(KID_BN_BWD_FINAPPLY, st, 24.0 * n * c);
        const BnFinApply A0{part, nblk, 4 * c, 0, x[0], gy[0], mean[0], rstd[0], gamma[0], beta[0], gx[0], dbeta[0], dgamma[0], nullptr,
                            nullptr, nullptr};
        const BnFinApply A1{part, nblk, 4 * c, 2 * c, x[1], gy[1], mean[1], rstd[1], gamma[1], beta[1], gx[1], dbeta[1], dgamma[1],
                            nullptr, nullptr, nullptr};
        launch_finapply(st, n, c, relu, training, false, 2, A0, A1);
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    launch_finalize(st, (const float *)part, nblk, 4 * c, MapBnPair{dbeta[0], dgamma[0], dbeta[1], dgamma[1], c});
    const long long total4 = (long long)n * (c >> 2);
    const int nb2 = (int)std::min<long long>((total4 + TPB - 1) / TPB, 256 * 16);
    {
        PtvScopedTimer t(KID_BN_BWD_APPLY, st, 24.0 * n * c);
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb2, 2), dim3(TPB), 0, st, total4, c >> 2, 1.0f / (float)n, x[0], gy[0],
                           mean[0], rstd[0], gamma[0], beta[0], relu, (const float *)dbeta[0], (const float *)dgamma[0], training,
                           gx[0], sec);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int skinny_linear_forward_xf_hip_launcher(int n, int cin, int cout, const float *x, const float *W,
                                                    const float *xsc, const float *xsh, float *y, void *stream) {
    if (n < 0 || cin < 4 || cin % 4 != 0 || cout < 1 || cout > 64 || (xsc == nullptr) != (xsh == nullptr)) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const size_t lds = sizeof(float) * ((size_t)cout * (cin + 4) + 2 * (size_t)cin);
    if (lds > 160 * 1024) return PTV2_ERR_ARG;
    if (lds > 32 * 1024)
        (void)hipFuncSetAttribute((const void *)skinny_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long long total = (long long)n * cout;
    const int nblk = (int)std::min<long long>((total + TPB - 1) / TPB, 256 * 8);
    {
        PtvScopedTimer t(KID_SKINNY_FWD, (hipStream_t)stream, 4.0 * n * (cin + cout));
        hipLaunchKernelGGL(skinny_fwd_kernel, dim3(nblk), dim3(TPB), lds, (hipStream_t)stream, (long long)n, cin, cout, x, W, xsc,
                           xsh, y, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (float *)nullptr);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// two projections through the same W in one launch (internal to the block runtime): y[i] = f(x[i]) W^T, i = 0, 1
int skinny_linear_forward_pair(int n, int cin, int cout, const float *const *x, const float *W, const float *const *xsc,
                               const float *const *xsh, float *const *y, void *stream) {
    if (n < 0 || cin < 4 || cin % 4 != 0 || cout < 1 || cout > 64) return PTV2_ERR_ARG;
    if ((xsc[0] == nullptr) != (xsc[1] == nullptr)) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const size_t lds = sizeof(float) * ((size_t)cout * (cin + 4) + 2 * (size_t)cin);
    if (lds > 160 * 1024) return PTV2_ERR_ARG;
    if (lds > 32 * 1024)
        (void)hipFuncSetAttribute((const void *)skinny_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long long total = (long long)n * cout;
    const int nblk = (int)std::min<long long>((total + TPB - 1) / TPB, 256 * 8);
    {
        PtvScopedTimer t(KID_SKINNY_FWD, (hipStream_t)stream, 8.0 * n * (cin + cout));
        hipLaunchKernelGGL(skinny_fwd_kernel, dim3(nblk, 2), dim3(TPB), lds, (hipStream_t)stream, (long long)n, cin, cout, x[0], W,
                           xsc[0], xsh[0], y[0], x[1], xsc[1], xsh[1], y[1]);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int skinny_linear_forward_hip_launcher(int n, int cin, int cout, const float *x, const float *W, float *y,
                                                 void *stream) {
    return skinny_linear_forward_xf_hip_launcher(n, cin, cout, x, W, nullptr, nullptr, y, stream);
}

extern "C" int skinny_linear_backward_hip_launcher(int n, int cin, int cout, const float *gy, const float *W, float *gx,
                                                  void *stream) {
    if (n < 0 || cin < 4 || cin % 4 != 0 || cout < 1) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const long long total = (long long)n * (cin >> 2);
    const int nblk = (int)std::min<long long>((total + TPB - 1) / TPB, 256 * 16);
    {
        PtvScopedTimer t(KID_SKINNY_BWD, (hipStream_t)stream, 4.0 * n * (cin + cout));
        hipLaunchKernelGGL(skinny_bwd_kernel, dim3(nblk), dim3(TPB), 0, (hipStream_t)stream, (long long)n, cin, cout, gy, W, gx,
                           (const float *)nullptr, (float *)nullptr, nblk, gva::PtvRiders{});
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// gx[i] = gy[i] W for two gradient tensors in one launch (internal to the block runtime)
int skinny_linear_backward_pair(int n, int cin, int cout, const float *const *gy, const float *W, float *const *gx, void *stream) {
    if (n < 0 || cin < 4 || cin % 4 != 0 || cout < 1) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const long long total = (long long)n * (cin >> 2);
    const int nblk = (int)std::min<long long>((total + TPB - 1) / TPB, 256 * 16);
    {
        PtvScopedTimer t(KID_SKINNY_BWD, (hipStream_t)stream, 8.0 * n * (cin + cout));
        // the parameter-gradient sums queued by the stages before (logits parameters, kW / qW weights) ride along: gx
        // depends on none of them
        const gva::PtvRiders Rs = gva::ptv2_rider_take();
        hipLaunchKernelGGL(skinny_bwd_kernel, dim3(nblk + gva::rider_blocks(Rs), 2), dim3(TPB), 0, (hipStream_t)stream, (long long)n,
                           cin, cout, gy[0], W, gx[0], gy[1], gx[1], nblk, Rs);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// ---- deferred weight-gradient launches (see WgradJob) ----------------------------------------------------------------------
// gva_wgrad_tile.hip
int gva_wgrad_tile_supported(int k, int c, int g);
size_t gva_wgrad_tile_plan(dense::WgradJob *J, int max_splits);
int gva_wgrad_tile_launch_one(const dense::WgradJob &J, hipStream_t st);
int gva_wgrad_tile_launch_jobs(const dense::WgradJob *table, int njobs, int wgs, int pos_wgs, hipStream_t st);
namespace {
constexpr int WGRAD_FORMS = 6;
struct WgradDefer {
    bool active = false;
    bool armed = false;          // the call in progress may be filed (set by the call sites whose operands outlive their Block)
    bool armed_rs = false;       // ... the row-scaled strided form (the grouped projection's weight gradient inside the attention)
    char *arena = nullptr;       // [job table RS = 0 | job table RS = 1 | kept operands and chunk records]
    size_t cap = 0, used = 0;
    // filed since the last flush, per kernel form: LDS-staged (RS = 0 / 1), direct fp32, direct bf16, grouped, grouped with A recomputed
    std::vector<WgradJob> jobs[WGRAD_FORMS];
    double bytes[WGRAD_FORMS] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // their algorithmic bytes (kernel timer)
};
thread_local WgradDefer g_wdefer;
constexpr int WGRAD_MAX_JOBS = 64;
constexpr size_t WGRAD_TABLE_BYTES = (sizeof(WgradJob) * WGRAD_MAX_JOBS + 255) & ~(size_t)255;
}  // namespace

void ptv2_wgrad_defer_begin(void *arena, size_t bytes) {
    WgradDefer &D = g_wdefer;
    for (int f = 0; f < WGRAD_FORMS; ++f) { D.jobs[f].clear(); D.bytes[f] = 0.0; }
    D.armed = D.armed_rs = false;
    D.active = arena != nullptr && bytes > WGRAD_FORMS * WGRAD_TABLE_BYTES;
    D.arena = (char *)arena;
    D.cap = bytes;
    D.used = WGRAD_FORMS * WGRAD_TABLE_BYTES;
}
bool ptv2_wgrad_defer_active() { return g_wdefer.active; }
void ptv2_wgrad_defer_end() {
    g_wdefer.active = g_wdefer.armed = g_wdefer.armed_rs = false;
    for (auto &j : g_wdefer.jobs) j.clear();
}
void ptv2_wgrad_defer_arm(bool on) { g_wdefer.armed = on && g_wdefer.active; }
void ptv2_wgrad_defer_arm_rs(bool on) { g_wdefer.armed_rs = on && g_wdefer.active; }
bool ptv2_wgrad_defer_armed_rs() { return g_wdefer.active && g_wdefer.armed_rs; }
size_t ptv2_wgrad_defer_table_bytes() { return WGRAD_FORMS * WGRAD_TABLE_BYTES; }
// a slice of the arena that lives until the backward ends (operands a deferred job reads, its records); NULL: no room
float *ptv2_wgrad_defer_alloc(size_t floats) {
    WgradDefer &D = g_wdefer;
    const size_t bytes = (sizeof(float) * floats + 255) & ~(size_t)255;
    if (!D.active || D.used + bytes > D.cap) return nullptr;
    float *p = (float *)(D.arena + D.used);
    D.used += bytes;
    return p;
}
// runs the jobs filed so far: per kernel form the table writers, the batched kernel, the batched finalize
int ptv2_wgrad_defer_flush(void *stream) {
    WgradDefer &D = g_wdefer;
    if (!D.active) return PTV2_OK;
    hipStream_t st = (hipStream_t)stream;
    static const bool once = [] {
        return hipFuncSetAttribute((const void *)linear_wgrad_lds_kernel_jobs<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)WL_LDS_BYTES) == hipSuccess &&
               hipFuncSetAttribute((const void *)linear_wgrad_lds_kernel_jobs<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)WL_LDS_BYTES) == hipSuccess;
    }();
    if (!once) return PTV2_ERR_LAUNCH;
    for (int form = 0; form < WGRAD_FORMS; ++form) {
        std::vector<WgradJob> &jobs = D.jobs[form];
        if (jobs.empty()) continue;
        WgradJob *table = (WgradJob *)(D.arena + (size_t)form * WGRAD_TABLE_BYTES);
        const int njobs = (int)jobs.size();
        // the jobs whose workgroups run longest first (rows per workgroup x the row piece it reads): the backward files the
        // full-resolution patch-embedding Block LAST, and its 150 us workgroups starting at the end of the launch were its tail
        // (bench step 10.53 -> 10.49 ms)
        std::stable_sort(jobs.begin(), jobs.end(), [form](const WgradJob &a, const WgradJob &b) {
            const long long wa = (long long)a.chunk * (form >= 4 ? a.gw * a.cin : 1), wb = (long long)b.chunk * (form >= 4 ? b.gw * b.cin : 1);
            return wa > wb;
        });
        int wgs = 0, fin = 0;
        long long pos_wgs = 0;  // (recompute form: workgroups of the relative-position launch in front of the jobs)
        bool pos_all = form == 5;
        for (WgradJob &J : jobs) {
            if (form == 5) {
                J.ldy = pos_wgs;
                pos_wgs += ((long long)J.n * 16 + 255) / 256;
                pos_all = pos_all && J.mX[0] != nullptr;
            }
            J.wg0 = wgs; J.fin0 = fin;
            J.fin_lanes = J.chunks > 32 ? 4 : 1;
            wgs += J.wgs;
            fin += J.fin_lanes == 1 ? (J.rec + 255) / 256 * 256 : (J.rec + 63) / 64 * 256;  // whole workgroups
        }
        for (int at = 0; at < njobs; at += WGRAD_PACK) {
            WgradJobPack pack;
            const int cnt = std::min(WGRAD_PACK, njobs - at);
            for (int i = 0; i < WGRAD_PACK; ++i) pack.j[i] = jobs[(size_t)std::min(at + i, njobs - 1)];
            hipLaunchKernelGGL(wgrad_jobs_write_kernel, dim3(1), dim3(64), 0, st, pack, cnt, table + at);
        }
        {
            PtvScopedTimer t(form == 5 ? KID_WGRAD_TILE : form == 4 ? KID_WGRAD_GROUPED : (form < 2 ? KID_WGRAD_LDS : KID_WGRAD), st, D.bytes[form]);
            if (form == 0)
                hipLaunchKernelGGL(linear_wgrad_lds_kernel_jobs<0>, dim3((unsigned)wgs), dim3(TPB), WL_LDS_BYTES, st,
                                   (const WgradJob *)table, njobs);
            else if (form == 1)
                hipLaunchKernelGGL(linear_wgrad_lds_kernel_jobs<1>, dim3((unsigned)wgs), dim3(TPB), WL_LDS_BYTES, st,
                                   (const WgradJob *)table, njobs);
            else if (form == 2)
                hipLaunchKernelGGL(linear_wgrad_kernel_jobs<false>, dim3((unsigned)wgs), dim3(TPB), 0, st, (const WgradJob *)table,
                                   njobs);
            else if (form == 3)
                hipLaunchKernelGGL(linear_wgrad_kernel_jobs<true>, dim3((unsigned)wgs), dim3(TPB), 0, st, (const WgradJob *)table,
                                   njobs);
            else if (form == 5) {
                if (gva_wgrad_tile_launch_jobs((const WgradJob *)table, njobs, wgs, pos_all ? (int)pos_wgs : 0, st) != PTV2_OK) return PTV2_ERR_LAUNCH;
            } else {
                size_t lds = 0;
                for (const WgradJob &J : jobs) lds = std::max(lds, grouped_lds_bytes(J.cin, J.gw));
                hipLaunchKernelGGL(grouped_wgrad_kernel_jobs, dim3((unsigned)wgs), dim3(TPB), lds, st, (const WgradJob *)table, njobs);
            }
        }
        hipLaunchKernelGGL(wgrad_jobs_finalize_kernel, dim3((unsigned)((fin + 255) / 256)), dim3(256), 0, st,
                           (const WgradJob *)table, njobs, fin);
        jobs.clear();
        D.bytes[form] = 0.0;
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// internal (gva_block.hip): the grouped projection's weight gradient with A recomputed from the saved softmax weights
// (gva_wgrad_tile.hip): dW (g, 8, c) and db (g, 8) = sum_n g_out sw.  Filed when the caller's backward defers (the operands
// outlive the Block), else launched here with its finalize; PTV2_ERR_ARG for shapes without an instance
int gva_wp2_wgrad_recompute(int n, int k, int c, int g, const float *g_out, const float *w, const float *sw, const float *a,
                            const float *b, const float *coord, const int *idx, float *dW, float *db, void *workspace,
                            size_t workspace_bytes, void *stream) {
    if (!gva_wgrad_tile_supported(k, c, g) || n < 1 || !g_out || !w || !sw || !a || !b || !coord || !idx || !dW || !db) return PTV2_ERR_ARG;
    WgradJob J{};
    J.n = n; J.cin = c; J.batch = g;
    J.gY = g_out; J.X = w; J.rowscale = sw; J.dW = dW; J.db = db;
    J.aux[0] = coord; J.aux[1] = idx; J.aux[2] = a; J.aux[3] = b;
    const double algo = 4.0 * ((double)n * (c + 16.0 * g + g + 16 + 3) + (double)c * c + c);  // g_out, w, sw, idx, coord in; dW, db out
    if (g_wdefer.active && g_wdefer.armed_rs && (int)g_wdefer.jobs[5].size() < WGRAD_MAX_JOBS) {
        const size_t floats = gva_wgrad_tile_plan(&J, n / 128 + 1);
        float *keep = ptv2_wgrad_defer_alloc(floats);
        float *pos = keep ? ptv2_wgrad_defer_alloc((size_t)n * 16 * 4) : nullptr;  // relative positions, written at the flush
        if (keep && pos) {
            J.part = keep;
            J.mX[0] = pos;
            g_wdefer.jobs[5].push_back(J);
            g_wdefer.bytes[5] += algo;
            return PTV2_OK;
        }
    }
    const size_t rec = (size_t)g * (8 * (size_t)c + 8);
    const int fit = (int)std::min<size_t>(1 << 20, workspace_bytes / (sizeof(float) * rec));
    if (!workspace || fit < 1) return PTV2_ERR_WORKSPACE;
    (void)gva_wgrad_tile_plan(&J, std::min(fit, n / 128 + 1));  // (the same split as a filed job: the same bits either way)
    J.part = (float *)workspace;
    hipStream_t st = (hipStream_t)stream;
    {
        PtvScopedTimer t(KID_WGRAD_TILE, st, algo);
        if (gva_wgrad_tile_launch_one(J, st) != PTV2_OK) return PTV2_ERR_LAUNCH;
    }
    launch_finalize(st, (const float *)J.part, J.chunks, (int)rec, gva::MapSplit2<float>{dW, db, g * 8 * c});
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// internal (gva_block.hip): rowscale != NULL asks for db[b][o] = sum_n gY[n, b, o] * rowscale[n * lds_s + b] instead of the plain
// column sums.  Only the LDS-staged fp32 kernel forms them: *weighted says whether it did (1) or whether the caller has to
// compute db itself (0: db is then not written at all).
extern "C" int linear_wgrad_strided_rowscale(int n, int cout, int cin, int batch, const float *gY, long long ldy, long long sy,
                                             const float *X, long long ldx, long long sx, float *dW, float *db,
                                             const float *rowscale, long long lds_s, int *weighted, void *workspace,
                                             size_t workspace_bytes, void *stream);

extern "C" int linear_wgrad_strided_hip_launcher(int n, int cout, int cin, int batch, const float *gY, long long ldy,
                                                 long long sy, const float *X, long long ldx, long long sx, float *dW,
                                                 float *db, void *workspace, size_t workspace_bytes, void *stream) {
    return linear_wgrad_strided_rowscale(n, cout, cin, batch, gY, ldy, sy, X, ldx, sx, dW, db, nullptr, 0, nullptr, workspace,
                                         workspace_bytes, stream);
}

extern "C" int linear_wgrad_strided_rowscale(int n, int cout, int cin, int batch, const float *gY, long long ldy, long long sy,
                                             const float *X, long long ldx, long long sx, float *dW, float *db,
                                             const float *rowscale, long long lds_s, int *weighted, void *workspace,
                                             size_t workspace_bytes, void *stream) {
    if (weighted) *weighted = 0;
    if (n < 1 || cout < 1 || cin < 1 || batch < 1) return PTV2_ERR_ARG;
    // the grouped projection's shape (eight output rows per group, operands as the attention backward passes them): the
    // vector-ALU kernel that reads whole row pieces of X (grouped_wgrad_tile); AO_AMD_WP2_GROUPED=0: the strided matrix-core form
    static const bool grouped_on = [] { const char *e = getenv("AO_AMD_WP2_GROUPED"); return !(e && e[0] == '0'); }();
    // (also when the matrix products run on bf16 operands: this one is a vector-ALU kernel, exact fp32 either way)
    if (grouped_on && rowscale && db && cout == GRP_I && cin % 4 == 0 && cin / 4 <= TPB &&
        ldy == (long long)batch * cout && sy == cout && ldx == (long long)batch * cin && sx == cin && lds_s == batch &&
        wgrad_lds_ok(gY, ldy, sy, X, ldx, sx)) {
        const int q = cin / 4;
        int gw = 1;
        for (int d = 1; d <= batch; ++d)
            if (batch % d == 0 && d * q <= TPB) gw = d;
        const int blocks_g = batch / gw;
        const int chunks = (int)std::max<long long>(1, std::min<long long>(((long long)n + 127) / 128, std::max(1, 768 / blocks_g)));
        const int chunk = (n + chunks - 1) / chunks;
        const size_t rec = (size_t)batch * ((size_t)cout * cin + cout);
        WgradJob J{};
        J.n = n; J.cout = cout; J.cin = cin; J.tiles_i = 1; J.tiles = blocks_g; J.batch = batch; J.chunk = chunk;
        J.chunks = (n + chunk - 1) / chunk; J.has_pb = 1; J.count = 0; J.rec = (int)rec; J.gw = gw;
        J.wgs = J.chunks * blocks_g;
        J.ldy = ldy; J.sy = sy; J.ldx = ldx; J.sx = sx; J.lds_s = lds_s;
        J.gY = gY; J.X = X; J.rowscale = rowscale; J.dW = dW; J.db = db;
        const double algo = 4.0 * batch * ((double)n * (cout + cin) + (double)cout * cin + cout + (double)n);
        if (g_wdefer.active && g_wdefer.armed_rs && (int)g_wdefer.jobs[4].size() < WGRAD_MAX_JOBS) {
            float *keep = ptv2_wgrad_defer_alloc((size_t)J.chunks * rec);
            if (keep) {
                J.part = keep;
                g_wdefer.jobs[4].push_back(J);
                g_wdefer.bytes[4] += algo;
                if (weighted) *weighted = 1;
                return PTV2_OK;
            }
        }
        if (workspace && workspace_bytes >= sizeof(float) * (size_t)J.chunks * rec) {
            hipStream_t st = (hipStream_t)stream;
            J.part = (float *)workspace;
            const size_t lds = grouped_lds_bytes(cin, gw);
            {
                PtvScopedTimer t(KID_WGRAD_GROUPED, st, algo);
                hipLaunchKernelGGL(grouped_wgrad_kernel, dim3((unsigned)J.wgs), dim3(TPB), lds, st, J);
            }
            launch_finalize(st, (const float *)J.part, J.chunks, (int)rec, gva::MapSplit2<float>{dW, db, batch * cout * cin});
            if (weighted) *weighted = 1;
            PTV2_CHECK_LAUNCH();
            return PTV2_OK;
        }
    }
    const int tiles_o = (cout + WG_TILE - 1) / WG_TILE, tiles_i = (cin + WG_TILE - 1) / WG_TILE;
    const int chunk = wg_chunk(n, tiles_o * tiles_i * batch, g_wdefer.active && (rowscale ? g_wdefer.armed_rs : g_wdefer.armed));
    const int chunks = (n + chunk - 1) / chunk;
    const size_t need = sizeof(float) * (size_t)chunks * batch * ((size_t)cout * cin + cout);
    if (!workspace || workspace_bytes < need) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    float *part_b = part;  // non-null flag: bias partials live behind the weight partials of each chunk record
    dim3 grid(chunks, tiles_o * tiles_i, batch);
    {
        // algorithmic bytes, strict: every operand read once, every result written once (the split-K partial
        // records of this implementation are its own overhead, not the op's)
        const bool use_lds = !ptv2_matmul_bf16() && wgrad_lds_shape_ok(cout, cin) && wgrad_lds_ok(gY, ldy, sy, X, ldx, sx);
        const bool rs = rowscale && db && use_lds;
        if (rowscale && !rs) db = nullptr;  // (the caller forms the weighted sums itself)
        if (rs && weighted) *weighted = 1;
        const int form = ptv2_matmul_bf16() ? 3 : (use_lds ? 0 : 2);
        // (also the row-scaled call whose weighted bias sums this path cannot form -- bf16 operands: the caller computes them
        // itself and has kept gY for the deferral)
        const bool plain = (!rowscale && g_wdefer.armed) || (rowscale && !rs && g_wdefer.armed_rs);
        if (plain && g_wdefer.active && (int)g_wdefer.jobs[form].size() < WGRAD_MAX_JOBS) {
            // (the plain strided form, armed by a caller that keeps gY alive: the Linear + BatchNorm layers between the stages)
            const size_t rec = (size_t)batch * ((size_t)cout * cin + (db ? cout : 0));
            float *keep = ptv2_wgrad_defer_alloc((size_t)chunks * rec);
            if (keep) {
                WgradJob J{};
                J.n = n; J.cout = cout; J.cin = cin; J.tiles_i = tiles_i; J.tiles = tiles_o * tiles_i; J.batch = batch;
                J.chunk = chunk; J.chunks = chunks; J.has_pb = db ? 1 : 0; J.count = 0; J.rec = (int)rec;
                J.wgs = chunks * J.tiles * J.batch;
                J.ldy = ldy; J.sy = sy; J.ldx = ldx; J.sx = sx;
                J.gY = gY; J.X = X; J.part = keep; J.dW = dW; J.db = db;
                g_wdefer.jobs[form].push_back(J);
                g_wdefer.bytes[form] += 4.0 * batch * ((double)n * (cout + cin) + (double)cout * cin + (db ? cout : 0));
                return PTV2_OK;
            }
        }
        if (rs && g_wdefer.active && g_wdefer.armed_rs && (int)g_wdefer.jobs[1].size() < WGRAD_MAX_JOBS) {
            // inside a model backward (the caller keeps gY alive until its end): filed, run with the other Blocks' (WgradJob)
            const size_t rec = (size_t)batch * ((size_t)cout * cin + cout);
            float *keep = ptv2_wgrad_defer_alloc((size_t)chunks * rec);
            if (keep) {
                WgradJob J{};
                J.n = n; J.cout = cout; J.cin = cin; J.tiles_i = tiles_i; J.tiles = tiles_o * tiles_i; J.batch = batch;
                J.chunk = chunk; J.chunks = chunks; J.has_pb = 1; J.count = 0; J.rec = (int)rec;
                J.wgs = chunks * J.tiles * J.batch;
                J.ldy = ldy; J.sy = sy; J.ldx = ldx; J.sx = sx; J.lds_s = lds_s;
                J.gY = gY; J.X = X; J.rowscale = rowscale; J.part = keep; J.dW = dW; J.db = db;
                g_wdefer.jobs[1].push_back(J);
                g_wdefer.bytes[1] += 4.0 * batch * ((double)n * (cout + cin) + (double)cout * cin + cout + (double)n);
                return PTV2_OK;
            }
        }
        PtvScopedTimer t(use_lds ? KID_WGRAD_LDS : KID_WGRAD, st,
                         4.0 * batch * ((double)n * (cout + cin) + (double)cout * cin + (db ? cout : 0) + (rs ? (double)n : 0.0)));
        if (ptv2_matmul_bf16())
            hipLaunchKernelGGL(linear_wgrad_kernel<true>, grid, dim3(TPB), 0, st, n, cout, cin, tiles_i, gY, ldy, sy, X, ldx, sx, part,
                               db ? part_b : (float *)nullptr, batch, WgradMulti{}, chunk);
        else if (use_lds) {
            static const bool once = [] {
                return hipFuncSetAttribute((const void *)linear_wgrad_lds_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)WL_LDS_BYTES) == hipSuccess;
            }();
            (void)once;
            if (rs) {
                static const bool once1 = [] {
                    return hipFuncSetAttribute((const void *)linear_wgrad_lds_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)WL_LDS_BYTES) == hipSuccess;
                }();
                (void)once1;
                hipLaunchKernelGGL(linear_wgrad_lds_kernel<1>, grid, dim3(TPB), WL_LDS_BYTES, st, n, cout, cin, tiles_i, gY, ldy, sy, X,
                                   ldx, sx, part, part_b, batch, WgradMulti{}, chunk, rowscale, lds_s);
            } else
                hipLaunchKernelGGL(linear_wgrad_lds_kernel<0>, grid, dim3(TPB), WL_LDS_BYTES, st, n, cout, cin, tiles_i, gY, ldy, sy, X,
                                   ldx, sx, part, db ? part_b : (float *)nullptr, batch, WgradMulti{}, chunk, (const float *)nullptr, 0LL);
        } else
            hipLaunchKernelGGL(linear_wgrad_kernel<false>, grid, dim3(TPB), 0, st, n, cout, cin, tiles_i, gY, ldy, sy, X, ldx, sx, part,
                               db ? part_b : (float *)nullptr, batch, WgradMulti{}, chunk);
    }
    if (db) launch_finalize(st, (const float *)part, chunks, batch * cout * cin + batch * cout,
                            gva::MapSplit2<float>{dW, db, batch * cout * cin});
    else launch_finalize(st, (const float *)part, chunks, batch * cout * cin, gva::MapVec<float>{dW});
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

struct MapWgradMulti {  // record = [count][cout*cin] weights, then [count][cout] bias sums
    WgradMulti m;
    int wlen, cout;
    __device__ void operator()(int e, double v) const {
        const int wtot = m.count * wlen;
        if (e < wtot) {
            const int b = e / wlen;
            m.dW[b][e - b * wlen] = (float)v;
        } else {
            const int r = e - wtot, b = r / cout;
            if (m.db[b]) m.db[b][r - b * cout] = (float)v;
        }
    }
};


namespace gva {
template <> struct RiderOf<MapWgradMulti> {
    static constexpr bool ok = true;
    static PtvRider make(const MapWgradMulti &m) {
        PtvRider r{};
        r.kind = RIDER_WGRADN;
        for (int i = 0; i < 6; ++i) { r.p[i] = i < m.m.count ? m.m.dW[i] : nullptr; r.p[6 + i] = i < m.m.count ? m.m.db[i] : nullptr; }
        r.i0 = m.wlen; r.i1 = m.cout; r.i2 = m.m.count;
        return r;
    }
};
}  // namespace gva

// count (<= 6) products dW[i] (cout,cin) = gY[i]^T X[i], db[i] = column sums of gY[i] (db[i] may be NULL), all of one
// shape and row count, in one launch + one finalize (workspace: dense_workspace_bytes(n, count * cout, cin))
extern "C" int linear_wgrad_multi_hip_launcher(int n, int cout, int cin, int count, const float *const *gY,
                                               const float *const *X, float *const *dW, float *const *db,
                                               const float *const *xsc, const float *const *xsh, void *workspace,
                                               size_t workspace_bytes, void *stream) {
    if (n < 1 || cout < 1 || cin < 1 || count < 1 || count > 6 || !gY || !X || !dW) return PTV2_ERR_ARG;
    const int chunk = wg_chunk(n, ((cout + WG_TILE - 1) / WG_TILE) * ((cin + WG_TILE - 1) / WG_TILE) * count,
                               g_wdefer.active && g_wdefer.armed);
    const int chunks = (n + chunk - 1) / chunk;
    const size_t rec = (size_t)count * ((size_t)cout * cin + cout);
    if (!workspace || workspace_bytes < sizeof(float) * (size_t)chunks * rec) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    WgradMulti m{};
    m.count = count;
    for (int i = 0; i < count; ++i) {
        if (!gY[i] || !X[i] || !dW[i]) return PTV2_ERR_ARG;
        m.gY[i] = gY[i]; m.X[i] = X[i]; m.dW[i] = dW[i]; m.db[i] = db ? db[i] : nullptr;
        m.xsc[i] = xsc ? xsc[i] : nullptr;
        m.xsh[i] = xsh ? xsh[i] : nullptr;
        if ((m.xsc[i] == nullptr) != (m.xsh[i] == nullptr)) return PTV2_ERR_ARG;
    }
    float *part = (float *)workspace;
    const int tiles_o = (cout + WG_TILE - 1) / WG_TILE, tiles_i = (cin + WG_TILE - 1) / WG_TILE;
    dim3 grid(chunks, tiles_o * tiles_i, count);
    {
        // algorithmic bytes, strict: gY[i] once each, every DISTINCT X once (q, k, v share theirs), dW / db once each
        int distinct_x = 0;
        for (int i = 0; i < count; ++i) {
            bool seen = false;
            for (int j = 0; j < i; ++j) seen |= m.X[j] == m.X[i] && m.xsc[j] == m.xsc[i];
            distinct_x += !seen;
        }
        bool lds_ok = !ptv2_matmul_bf16() && wgrad_lds_shape_ok(cout, cin);
        for (int i = 0; i < count && lds_ok; ++i) lds_ok = wgrad_lds_ok(m.gY[i], cout, 0, m.X[i], cin, 0);
        const double algo_bytes = 4.0 * ((double)count * n * cout + (double)distinct_x * n * cin + (double)count * cout * (cin + 1));
        const int form = ptv2_matmul_bf16() ? 3 : (lds_ok ? 0 : 2);
        if (g_wdefer.active && g_wdefer.armed && (int)g_wdefer.jobs[form].size() < WGRAD_MAX_JOBS) {
            // inside a model backward: filed, and run with all the others by ONE launch at the end (WgradJob); the records go
            // to the arena (the caller's workspace is reused before that launch)
            float *keep = ptv2_wgrad_defer_alloc((size_t)chunks * rec);
            if (keep) {
                WgradJob J{};
                J.n = n; J.cout = cout; J.cin = cin; J.tiles_i = tiles_i; J.tiles = tiles_o * tiles_i; J.batch = count;
                J.chunk = chunk; J.chunks = chunks; J.has_pb = 1; J.count = count; J.rec = (int)rec;
                J.wgs = chunks * J.tiles * J.batch;
                J.ldy = cout; J.ldx = cin; J.part = keep;
                for (int i = 0; i < count; ++i) {
                    J.mgY[i] = m.gY[i]; J.mX[i] = m.X[i]; J.mxsc[i] = m.xsc[i]; J.mxsh[i] = m.xsh[i];
                    J.mdW[i] = m.dW[i]; J.mdb[i] = m.db[i];
                }
                g_wdefer.jobs[form].push_back(J);
                g_wdefer.bytes[form] += algo_bytes;
                return PTV2_OK;
            }
        }
        PtvScopedTimer t(lds_ok ? KID_WGRAD_LDS : KID_WGRAD, st, algo_bytes);
        if (ptv2_matmul_bf16())
            hipLaunchKernelGGL(linear_wgrad_kernel<true>, grid, dim3(TPB), 0, st, n, cout, cin, tiles_i, (const float *)nullptr,
                               (long long)cout, 0LL, (const float *)nullptr, (long long)cin, 0LL, part, part, count, m, chunk);
        else if (lds_ok) {
            static const bool once = [] {
                return hipFuncSetAttribute((const void *)linear_wgrad_lds_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)WL_LDS_BYTES) == hipSuccess;
            }();
            (void)once;
            hipLaunchKernelGGL(linear_wgrad_lds_kernel<0>, grid, dim3(TPB), WL_LDS_BYTES, st, n, cout, cin, tiles_i,
                               (const float *)nullptr, (long long)cout, 0LL, (const float *)nullptr, (long long)cin, 0LL, part, part,
                               count, m, chunk, (const float *)nullptr, 0LL);
        } else
            hipLaunchKernelGGL(linear_wgrad_kernel<false>, grid, dim3(TPB), 0, st, n, cout, cin, tiles_i, (const float *)nullptr,
                               (long long)cout, 0LL, (const float *)nullptr, (long long)cin, 0LL, part, part, count, m, chunk);
    }
    launch_finalize(st, (const float *)part, chunks, (int)rec, MapWgradMulti{m, cout * cin, cout});
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int linear_wgrad_hip_launcher(int n, int cout, int cin, const float *gY, const float *X, float *dW,
                                         float *db, void *workspace, size_t workspace_bytes, void *stream) {
    return linear_wgrad_strided_hip_launcher(n, cout, cin, 1, gY, cout, 0, X, cin, 0, dW, db, workspace, workspace_bytes,
                                             stream);
}
