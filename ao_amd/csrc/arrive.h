// ao_amd/csrc/arrive.h -- column sums over per-workgroup records, finished INSIDE the producing kernel (round 5).
//
// Every BatchNorm of the step (point_transformer_v2m2_base.py:26-45) ends a split reduction: the kernel that produces the
// rows leaves one record per 64-row block (forward: column sum + centred sum of squares; backward: sum g', sum g' xhat), and
// a separate 5-9 us launch merged them -- ~100 launches per step whose whole cost is a launch boundary on the critical
// path.  Here the producing workgroups finish the sum themselves, in two levels so that nobody walks more than ~100 records:
//   level 1: the records of a column block are cut into groups of `gs` consecutive row blocks; the workgroup of a group that
//            arrives LAST (one agent-scope counter per group) adds the group's records in record order (float64) and
//            leaves one float64 partial;  with one group it emits directly;
//   level 2: the workgroup that completes the last group of a column block adds the partials in group order and emits.
// The association is fixed by (gs, slices), never by who arrives last: bitwise reproducible.  Protocol of an arrival as in
// gva_common.h::last_block_arrives (records written with agent-scope stores, store queue drained, barrier, relaxed
// agent-scope counter bump; the last block invalidates and reads with plain loads; it zeroes the counter for the next launch).
#pragma once
#include "common.h"

namespace arrive {

constexpr int SINGLE_MAX = 96;   // up to this many records per column block: one level
constexpr int GROUP_MIN = 32;    // records per first-level group otherwise (raised so that groups <= GROUPS_MAX)
constexpr int GROUPS_MAX = 128;

struct Args {            // by value in the kernel arguments
    unsigned *counters;  // zeroed region of this launch: per column block (x set) [1 + ngroups]
    double *fold;        // [ngroups][2][ctot] level-1 partials (unused with one group)
    int nrec, gs, ngroups, ctot;
};

// ---- host ----
inline void plan(int nrec, int *gs, int *ngroups) {
    int g = nrec <= SINGLE_MAX ? (nrec > 0 ? nrec : 1) : GROUP_MIN;
    while ((nrec + g - 1) / g > GROUPS_MAX) g *= 2;
    *gs = g;
    *ngroups = nrec > 0 ? (nrec + g - 1) / g : 1;
}
__host__ __device__ inline size_t fold_doubles(int ctot) { return (size_t)GROUPS_MAX * 2 * (size_t)ctot; }
// counters of a launch with `colblocks` column blocks (x sets); ARRIVE_COUNTERS bounds it (common.h)
inline long long counters_needed(int nrec, long long colblocks) {
    int gs, ng;
    plan(nrec, &gs, &ng);
    return colblocks * (ng + 1);
}
// the launch's counter region and plan; returns false when the launch does not fit the per-stream region
inline bool make(Args *A, hipStream_t st, int nrec, long long colblocks, double *fold, int ctot) {
    if (counters_needed(nrec, colblocks) > ARRIVE_COUNTERS) return false;
    unsigned *c = ptv2_stream_counters(st);
    if (!c) return false;
    A->counters = c + CNT_ARRIVE;
    A->fold = fold;
    A->nrec = nrec;
    A->ctot = ctot;
    plan(nrec, &A->gs, &A->ngroups);
    return true;
}

#if defined(__HIPCC__)
// ---- device ----
__device__ __forceinline__ void store_f(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_f4(float *p, float4 v) {
    store_f(p, v.x); store_f(p + 1, v.y); store_f(p + 2, v.z); store_f(p + 3, v.w);
}
__device__ __forceinline__ void store_d(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every thread of the workgroup calls, after the workgroup's record stores; true (in all its threads) in exactly one of the
// `expected` workgroups that share `counter`
__device__ __forceinline__ bool arrive_at(unsigned *counter, unsigned expected, int *s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record stores are acknowledged before the arrival is published
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == expected - 1;
        *s_flag = last;
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
    }
    __syncthreads();
    const bool last = *s_flag != 0;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // records of the other XCDs: invalidate, then read
    return last;
}

// Column sums of records [r0, r1) for the columns [col0, col0 + ncols) by the calling workgroup (blockDim.x threads, a
// multiple of 64, <= 1024): thread -> (column, slice); a slice walks its records in order with four in flight, the slices are
// combined in slice order.  rec.get(r, col, x, y) loads the two values of record r at column col; rec.add(r, x, y, a, b)
// accumulates them.  done(col, a, b) runs in one thread per column.  lds: 2 * blockDim.x doubles.
template <class Rec, class Done>
__device__ __forceinline__ void sum_cols(int r0, int r1, int col0, int ncols, double *lds, Rec rec, Done done) {
    const int T = blockDim.x;
    rec.init();
    for (int j0 = 0; j0 < ncols; j0 += T) {
        const int cols = (ncols - j0) < T ? (ncols - j0) : T;
        const int SL = T / cols;
        const int cj = threadIdx.x % cols, sl = threadIdx.x / cols;
        const int col = col0 + j0 + cj;
        double a = 0.0, b = 0.0;
        if (sl < SL) {
            int r = r0 + sl;
            for (; r + 3 * SL < r1; r += 4 * SL) {
                typename Rec::V x[4], y[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rec.get(r + u * SL, col, x[u], y[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) rec.add(r + u * SL, x[u], y[u], a, b);
            }
            for (; r < r1; r += SL) {
                typename Rec::V x, y;
                rec.get(r, col, x, y);
                rec.add(r, x, y, a, b);
            }
        }
        __syncthreads();  // (lds may still be read by the previous pass / the caller)
        lds[threadIdx.x] = a;
        lds[T + threadIdx.x] = b;
        __syncthreads();
        if ((int)threadIdx.x < cols) {
            double ta = 0.0, tb = 0.0;
            for (int t = 0; t < SL; ++t) { ta += lds[t * cols + cj]; tb += lds[T + t * cols + cj]; }
            done(col, ta, tb);
        }
    }
}

struct FoldRec {  // level-1 partials
    typedef double V;
    const double *fold; int ctot;
    __device__ __forceinline__ void init() {}
    __device__ __forceinline__ void get(int r, int col, double &x, double &y) const {
        x = fold[(size_t)r * 2 * ctot + col];
        y = fold[(size_t)r * 2 * ctot + ctot + col];
    }
    __device__ __forceinline__ void add(int, double x, double y, double &a, double &b) const { a += x; b += y; }
};

// All threads of the workgroup call, after writing record `rec_id` of column block `cb` (columns [col0, col0 + ncols)) with
// store_f.  emit(col, a, b) runs once per column, in the workgroup that completes the column block.
// s_flag: one __shared__ int; lds: 2 * blockDim.x doubles, free to overwrite.
template <class Rec, class Emit>
__device__ __forceinline__ void finish(const Args &A, int rec_id, int cb, int col0, int ncols, double *lds, int *s_flag, Rec rec,
                                       Emit emit) {
    const int grp = rec_id / A.gs;
    const int r0 = grp * A.gs, r1 = (r0 + A.gs) < A.nrec ? (r0 + A.gs) : A.nrec;
    unsigned *c = A.counters + (size_t)cb * (A.ngroups + 1);
    if (!arrive_at(c + 1 + grp, (unsigned)(r1 - r0), s_flag)) return;
    if (A.ngroups == 1) {
        sum_cols(r0, r1, col0, ncols, lds, rec, emit);
        return;
    }
    double *f = A.fold + (size_t)grp * 2 * A.ctot;
    const int ctot = A.ctot;
    sum_cols(r0, r1, col0, ncols, lds, rec, [=](int col, double a, double b) { store_d(f + col, a); store_d(f + ctot + col, b); });
    if (!arrive_at(c, (unsigned)A.ngroups, s_flag)) return;
    sum_cols(0, A.ngroups, col0, ncols, lds, FoldRec{A.fold, A.ctot}, emit);
}
#endif

}  // namespace arrive

// ---- what the sums are for: BatchNorm forward statistics / backward parameter sums, emitted by the finishing workgroup ----
namespace bnfin {

struct Emit {  // destinations of one BatchNorm's forward statistics (NULL: not wanted)
    float *mean, *rstd, *sc, *sh, *run_mean, *run_var;
    long long *batches;
    const float *gamma, *beta;
};

#if defined(__HIPCC__)
// t1 = sum x, t2 = sum_b (M2_b + S_b^2 / n_b) over the n rows of a column (parallel-variance identity, float64).
// No float64 division or square root here (each costs the HOST kernel ~20 registers for a path one workgroup in hundreds
// takes): 1 / n comes in as float64 from the host, and 1 / sqrt(v) is the float32 estimate refined by one Newton step in
// float64 (relative error ~1e-14 before the rounding to float32).
struct Norm {  // of a BatchNorm over n rows (host): 1 / n; n / (n - 1) (1 for n = 1); the partial last 64-row record and 1 / its rows
    double inv_n, unbias, inv_last;
    int last;
};
inline Norm norm_of(int n) {
    const int last = n > 0 ? (n - 1) / 64 : 0;
    return Norm{1.0 / (double)(n > 0 ? n : 1), n > 1 ? (double)n / (double)(n - 1) : 1.0, 1.0 / (double)(n > 0 ? n - last * 64 : 1), last};
}
__device__ __forceinline__ void emit_stats(const Emit &S, int ch, double t1, double t2, const Norm N, float eps, float momentum) {
    const double m = t1 * N.inv_n;
    double var = t2 * N.inv_n - m * m;
    var = var > 0.0 ? var : 0.0;
    const double v = var + (double)eps;
    double y = (double)__builtin_amdgcn_rsqf((float)v);
    y = y * (1.5 - 0.5 * v * y * y);
    const float mf = (float)m, rf = (float)y;
    S.mean[ch] = mf;
    S.rstd[ch] = rf;
    if (S.sc) {  // y = x * sc + sh is the whole normalisation: consumers apply it on their operand load
        const float scale = rf * S.gamma[ch];
        S.sc[ch] = scale;
        S.sh[ch] = S.beta[ch] - mf * scale;
    }
    if (S.run_mean) {
        S.run_mean[ch] = (float)((1.0 - momentum) * (double)S.run_mean[ch] + momentum * m);
        S.run_var[ch] = (float)((1.0 - momentum) * (double)S.run_var[ch] + momentum * (var * N.unbias));
        if (ch == 0 && S.batches) *S.batches += 1;
    }
}

// the same from sums of (x - x0), (x - x0)^2 (bn_stats_kernel: shifted by one sample of the column against cancellation)
__device__ __forceinline__ void emit_stats_shifted(const Emit &S, int ch, double t1, double t2, double x0, const Norm N, float eps,
                                                   float momentum) {
    const double d = t1 * N.inv_n, m = x0 + d;
    double var = t2 * N.inv_n - d * d;
    var = var > 0.0 ? var : 0.0;
    const double v = var + (double)eps;
    double y = (double)__builtin_amdgcn_rsqf((float)v);
    y = y * (1.5 - 0.5 * v * y * y);
    const float mf = (float)m, rf = (float)y;
    S.mean[ch] = mf;
    S.rstd[ch] = rf;
    if (S.sc) {
        const float scale = rf * S.gamma[ch];
        S.sc[ch] = scale;
        S.sh[ch] = S.beta[ch] - mf * scale;
    }
    if (S.run_mean) {
        S.run_mean[ch] = (float)((1.0 - momentum) * (double)S.run_mean[ch] + momentum * m);
        S.run_var[ch] = (float)((1.0 - momentum) * (double)S.run_var[ch] + momentum * (var * N.unbias));
        if (ch == 0 && S.batches) *S.batches += 1;
    }
}

// records [nrb][2][c] of a row GEMM's epilogue: per 64-row block the column sum and the sum of squares about the block mean
struct TileRec {
    typedef float V;
    const float *part; int c, n;
    int last; double inv_last;  // the (possibly partial) last record and 1 / its row count (from the host; 1 / 64 is exact for the others)
    __device__ __forceinline__ void init() {}
    __device__ __forceinline__ void get(int r, int col, float &x, float &y) const {
        x = part[(size_t)r * 2 * c + col];
        y = part[(size_t)r * 2 * c + c + col];
    }
    __device__ __forceinline__ void add(int r, float x, float y, double &a, double &b) const {
        const double sb = (double)x;
        a += sb;
        b += (double)y + sb * sb * (r == last ? inv_last : 0.015625);
    }
};
// records [nrec][stride] with the two sums of a column at off + col and off + c + col: plain sums (BatchNorm backward)
struct SumRec {
    typedef float V;
    const float *part; int stride, off, c;
    __device__ __forceinline__ void init() {}
    __device__ __forceinline__ void get(int r, int col, float &x, float &y) const {
        x = part[(size_t)r * stride + off + col];
        y = part[(size_t)r * stride + off + c + col];
    }
    __device__ __forceinline__ void add(int, float x, float y, double &a, double &b) const { a += (double)x; b += (double)y; }
};
#endif

}  // namespace bnfin
