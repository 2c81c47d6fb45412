// ao_amd/csrc/gva_common.h -- shared pieces of the fused grouped-vector-attention kernels.
#pragma once
#include <algorithm>

#include "common.h"

namespace gva {

constexpr int TPB = 256;
constexpr int WPB = TPB / WAVE;  // waves per block

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

constexpr int MAX_BLOCKS = 256 * 8;   // grid cap of the row-parallel stages
constexpr int MAX_PARAM_BLOCKS = 256;
constexpr size_t FUSED_FINAL_MAX = 16384;  // floats of partial records up to which the last block sums them itself // grid cap of the channel-parallel parameter-gradient stage

// floats of per-block partial sums any stage may write (the workspace's first region)
inline size_t part_floats(int c, int g) {
    size_t logits_fwd = (size_t)MAX_BLOCKS * 2 * g;
    size_t logits_bwd = (size_t)MAX_BLOCKS * g + std::max((size_t)MAX_PARAM_BLOCKS * 24576, (size_t)64 * c * (g + 4));
    size_t agg_bwd = (size_t)MAX_BLOCKS * (3 * (size_t)g + (size_t)g * g + 4 * (size_t)c);
    size_t m = logits_fwd > logits_bwd ? logits_fwd : logits_bwd;
    m = m > agg_bwd ? m : agg_bwd;
    return m < 9 * (size_t)MAX_BLOCKS ? 9 * (size_t)MAX_BLOCKS : m;
}
inline size_t rows_offset_bytes(int c, int g) { return align_up(sizeof(float) * part_floats(c, g)); }

// masked relative position of neighbour slot (n, s)
struct Rel {
    float x, y, z;
    int src;  // neighbour index or -1
};

__device__ __forceinline__ Rel rel_pos(const float *__restrict__ coord, const int *__restrict__ idx, long long row,
                                       int n) {
    Rel r;
    r.src = idx[row];
    r.x = r.y = r.z = 0.f;
    if (r.src >= 0) {
        r.x = coord[3 * (long long)r.src] - coord[3 * (long long)n];
        r.y = coord[3 * (long long)r.src + 1] - coord[3 * (long long)n + 1];
        r.z = coord[3 * (long long)r.src + 2] - coord[3 * (long long)n + 2];
    }
    return r;
}

// P = ReLU(a . pos + b): the folded Linear(3,C) -> BatchNorm -> ReLU of linear_p_bias
__device__ __forceinline__ float pe_act(float ax, float ay, float az, float b, float px, float py, float pz) {
    return fmaxf(__builtin_fmaf(az, pz, __builtin_fmaf(ay, py, __builtin_fmaf(ax, px, b))), 0.f);
}

// Which group a register of a (g, s) tile holds.  An MFMA result tile leaves row 4 q + r in register r of lane quarter q, and
// that register is the contraction step r of the next product.  With group = row, G = 6 occupies quarter 0 fully and half
// of quarter 1: all four registers are live in every lane, three quarters of the lanes compute padding, and every
// contraction over the groups is four matrix instructions.  With the groups dealt round-robin over the quarters instead
// (row 4 q + r holds group q + 4 r) G <= 8 needs registers 0..1 only (G <= 12: 0..2): the softmax and its backward -- DPP
// ladders per register -- and every product that contracts over the groups do half the work.  The kernel works on row
// numbers ("virtual" groups) throughout; the parameter tables in LDS are laid out by row, and gof / vof translate where a
// row number meets global memory (W1, gW1, g_A, g_sw rows, the parameter-gradient records).
#ifndef GVA_BWD_PERM
#define GVA_BWD_PERM 1
#endif
template <int G>
struct GroupRows {
    static constexpr bool PERM = GVA_BWD_PERM && G <= 12;
    static constexpr int RN = PERM ? (G + 3) / 4 : 4;  // registers of a quarter that can hold a group
    __host__ __device__ static constexpr int gof(int v) {  // row -> group, -1 for padding
        return PERM ? (((v & 3) < RN && v < 16 && (v >> 2) + 4 * (v & 3) < G) ? (v >> 2) + 4 * (v & 3) : -1) : (v < G ? v : -1);
    }
    __host__ __device__ static constexpr int vof(int g) { return PERM ? 4 * (g & 3) + (g >> 2) : g; }  // group -> row
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// Fixed-order column sums of a [nblk][len] partial matrix: out(j) = sum_b part[b][j].
// 64 columns x 16 row slices per 1024-thread block; each thread adds its slice sequentially (coalesced
// across columns), the 16 slice sums are combined in slice order -> bitwise reproducible.
// `Map` receives (column, sum) and writes the value wherever the stage wants it.
constexpr int FIN_COLS = 64, FIN_SLICES = 16;

// COLS x (1024 / COLS) row slices per 1024-thread block.  COLS = 64: wide records; COLS = 16: narrow records with many
// blocks (a few hundred columns, hundreds to thousands of records: the 64-column form left such a sum to 2 - 8 workgroups
// whose threads each walked 20 - 130 records, 8 - 13 us on the critical path of every Block)
template <class Map, int COLS = FIN_COLS>
__global__ __launch_bounds__(1024) void finalize_kernel(const float *__restrict__ part, int nblk, int len, Map map) {
    constexpr int SLICES = 1024 / COLS;
    __shared__ double s_acc[SLICES][COLS];
    const int col = threadIdx.x & (COLS - 1), sl = threadIdx.x / COLS;
    const int j = blockIdx.x * COLS + col;
    // four independent chains per thread (fixed association: ((a0+a1)+(a2+a3))) keep the loads in flight
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (j < len) {
        int b = sl;
        // (eight loads in flight per trip, added in the order of the four-chain loop below: the same bits; with many records
        // -- 1 000-2 000 at the full resolution -- four per trip left the sum a chain of dependent L2 round trips)
        for (; b + 7 * SLICES < nblk; b += 8 * SLICES) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u * SLICES) * len + j];
            a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
            a0 += (double)v[4]; a1 += (double)v[5]; a2 += (double)v[6]; a3 += (double)v[7];
        }
        for (; b + 3 * SLICES < nblk; b += 4 * SLICES) {
            a0 += (double)part[(size_t)b * len + j];
            a1 += (double)part[(size_t)(b + SLICES) * len + j];
            a2 += (double)part[(size_t)(b + 2 * SLICES) * len + j];
            a3 += (double)part[(size_t)(b + 3 * SLICES) * len + j];
        }
        for (; b < nblk; b += SLICES) a0 += (double)part[(size_t)b * len + j];
    }
    const double acc = (a0 + a1) + (a2 + a3);
    s_acc[sl][col] = acc;
    __syncthreads();
    if (sl == 0 && j < len) {
        double v = 0.0;
#pragma unroll 8  // (fully unrolled at SLICES = 64 the loads filled the 128 registers a 1024-thread block allows and spilled)
        for (int t = 0; t < SLICES; ++t) v += s_acc[t][col];
        map(j, v);
    }
}

// ---- the same final reduction inside the producing kernel ("last block done") -----------------------------
// Small reductions (a few hundred columns, a few hundred blocks) do not deserve a launch of their own: at deep
// stages the finalize launches were ~360 of ~2000 launches per step, each 5-6 us of pure latency.  Blocks write
// their record with agent-scope stores (visible across the 8 XCDs' L2s), bump an agent-scope counter, and the
// block that arrives last sums all records in block order -- the same fixed order as finalize_kernel, so results
// stay bitwise reproducible and independent of which block happens to be last.
__device__ __forceinline__ void part_store(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// only after last_block_arrives() (whose acquire fence dropped stale cache lines): plain loads,
// so that the column sums pipeline instead of paying one memory round trip per record
__device__ __forceinline__ float part_load(const float *p) { return *p; }
// call after the block's part_store()s; true in exactly one block (the last to arrive), in all of its threads
__device__ __forceinline__ bool last_block_arrives(unsigned *counter) {
    __shared__ int s_last;
    // EVERY thread's record stores must have been acknowledged by memory before thread 0 bumps the counter.  The records are
    // agent-scope atomic stores (sc1: written through, no L2 write-back needed), so a drained store queue is all a release
    // needs here -- but a workgroup-scope release fence does not emit it: the ISA of rounds 1-3 went `global_store ... sc1;
    // s_barrier; global_atomic_add` with no s_waitcnt vmcnt(0) in between, i.e. the last-arriving block (on another XCD)
    // could read a record that was still in flight: last launch's value of that slot, or on a fresh process whatever the
    // workspace held (the once-in-67-runs NaN loss of the 2-rank test).  An agent-scope release fence
    // would be correct too, but adds a buffer_wbl2 (a write-back of the whole L2) per block.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == total - 1;
        if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
    }
    __syncthreads();
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // records of the other XCDs: invalidate, then read
    return s_last != 0;
}
// column sums of part[nblk][len] by the calling block (blockDim.x == 256): the 256 threads are split into
// S = 256 / len record slices per column (a column summed by one thread alone is a chain of nblk dependent
// cross-XCD loads: 40 us for 400 records); slice sums are combined in slice order -> fixed association
template <class Map>
__device__ __forceinline__ void finalize_columns(const float *part, int nblk, int len, Map map) {
    __shared__ double s_fin[256];
    const int nth = blockDim.x < 256 ? blockDim.x : 256;
    for (int j0 = 0; j0 < len; j0 += nth) {  // one pass when len <= 256
        const int cols = (len - j0) < nth ? (len - j0) : nth;
        const int S = nth / cols;            // slices per column (>= 1)
        const int col = threadIdx.x % cols, sl = threadIdx.x / cols;
        double a0 = 0.0, a1 = 0.0;
        if ((int)threadIdx.x < cols * S) {
            int b = sl;
            for (; b + S < nblk; b += 2 * S) {
                a0 += (double)part_load(part + (size_t)b * len + j0 + col);
                a1 += (double)part_load(part + (size_t)(b + S) * len + j0 + col);
            }
            if (b < nblk) a0 += (double)part_load(part + (size_t)b * len + j0 + col);
        }
        __syncthreads();
        if (threadIdx.x < 256) s_fin[threadIdx.x] = a0 + a1;
        __syncthreads();
        if ((int)threadIdx.x < cols) {
            double v = 0.0;
            for (int t = 0; t < S; ++t) v += s_fin[t * cols + threadIdx.x];
            map(j0 + threadIdx.x, v);
        }
    }
}

// ---- riders --------------------------------------------------------------------------------------------------------
// A finalize whose result only a LATER stage needs (parameter gradients: nothing on the backward's critical path reads
// them) does not deserve a launch boundary of its own on that path (~5 us each, 5 per attention block).  Inside a
// PtvDeferScope launch_finalize() queues such a sum instead of launching it; the next designated host launch (one that
// neither reads its outputs nor overwrites its records) takes the queue and appends workgroups that run it beside its
// own work.  ptv2_rider_flush() launches whatever is still queued as a kernel of its own.
enum { RIDER_NONE = 0, RIDER_VEC, RIDER_LOGITS_PARAMS, RIDER_BWD_POINT, RIDER_WGRADN, RIDER_LOGITS_FUSED, RIDER_SPLIT2 };
struct PtvRider {
    const float *part;
    int nblk, len, kind, blocks;
    float *p[12];
    int i0, i1, i2;
};
constexpr int RIDER_QUEUE = 4;
struct PtvRiders {
    int count;
    PtvRider r[RIDER_QUEUE];
};
template <class Map> struct RiderOf { static constexpr bool ok = false; static PtvRider make(const Map &) { return PtvRider{}; } };
// host side (abi.hip; per host thread)
bool ptv2_rider_defer_active();
void ptv2_rider_defer_depth(int delta);
void ptv2_rider_defer(const PtvRider &r, hipStream_t st);  // queue of RIDER_QUEUE: one more flushes the oldest as its own launch
PtvRiders ptv2_rider_take();                               // the pending sums, for a host launch to carry
void ptv2_rider_flush(hipStream_t st);                     // whatever is pending, as launches of their own
int ptv2_rider_drop();                                     // forget whatever is pending (returns how many there were)
// A function that opens PtvDeferScopes holds one of these: queued riders point into THIS call's workspace records and
// gradient slots, so an early (error) return must not leave them for the next, unrelated host launch of the thread to run
struct PtvRiderGuard {
    bool armed = true;
    PtvRiderGuard() { (void)ptv2_rider_drop(); }  // the queue is empty on entry by construction; a stale entry is dropped
    ~PtvRiderGuard() { if (armed) (void)ptv2_rider_drop(); }
    void release() { armed = false; }
};
struct PtvDeferScope {
    PtvDeferScope() { ptv2_rider_defer_depth(1); }
    ~PtvDeferScope() { ptv2_rider_defer_depth(-1); }
};

// ---- attention dropout (GroupedVectorAttention.attn_drop, point_transformer_v2m2_base.py:101,122) inside the fused kernels.
// nn.Dropout(p) on the softmax output multiplies every (point, slot, group) weight by Bernoulli(1 - p) / (1 - p).  The mask
// is never stored: element e = (point * k + slot) * G + group of a Block's call gets the factor drop_factor(seed, e) -- a
// 32-bit integer hash against a threshold -- which the forward softmax kernels and the backward point kernel evaluate
// again (ao_amd/ptv2/gva.py::attn_drop_mask is the same function in torch, for the parity tests and the unfused path).
// thresh == 0: no dropout (factor 1).
struct PtvDrop { float scale; unsigned thresh, seed; };
__device__ __forceinline__ float ptv2_drop_factor(const PtvDrop d, unsigned long long e) {
    unsigned h = (unsigned)e ^ ((unsigned)(e >> 32) * 0x27D4EB2Fu);
    h = h * 0x9E3779B1u + d.seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h >= d.thresh ? d.scale : 0.f;
}
// host side (abi.hip; per host thread): the setting of the gva_block call in progress, read by the launchers of the
// softmax stage (forward) and of the point kernel (backward)
PtvDrop ptv2_attn_drop_current();
void ptv2_attn_drop_set(float p, unsigned seed);
struct PtvAttnDropScope {
    PtvDrop prev;
    PtvAttnDropScope(float p, unsigned seed) : prev(ptv2_attn_drop_current()) { ptv2_attn_drop_set(p, seed); }
    ~PtvAttnDropScope();
};

// few records, many columns (split-K weight gradients at the deep levels: 5-9 chunk records of 10^5 - 10^6 outputs):
// one thread per column walks the records -- the sliced form above would launch 16 threads per column of which at most
// nblk load anything (11 550 workgroups of 1 024 threads for 737 k columns: 24 us; this form: 2 880 of 256)
template <class Map>
__global__ __launch_bounds__(256) void finalize_flat_kernel(const float *__restrict__ part, int nblk, int len, Map map) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= len) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int b = 0;
    for (; b + 3 < nblk; b += 4) {
        a0 += (double)part[(size_t)b * len + j];
        a1 += (double)part[(size_t)(b + 1) * len + j];
        a2 += (double)part[(size_t)(b + 2) * len + j];
        a3 += (double)part[(size_t)(b + 3) * len + j];
    }
    for (; b < nblk; ++b) a0 += (double)part[(size_t)b * len + j];
    map(j, (a0 + a1) + (a2 + a3));
}

template <class Map>
inline void launch_finalize(hipStream_t st, const float *part, int nblk, int len, Map map) {
    if (RiderOf<Map>::ok && ptv2_rider_defer_active()) {  // the caller lets an independent later launch carry this sum
        PtvRider r = RiderOf<Map>::make(map);
        if (r.kind != RIDER_NONE) {
            r.part = part; r.nblk = nblk; r.len = len;
            r.blocks = nblk <= 32 ? (len + 255) / 256 : (nblk >= 1024 ? (len + 7) / 8 : (len + 15) / 16);  // rider_columns
            ptv2_rider_defer(r, st);
            return;
        }
    }
    if (nblk <= 32 && len >= 16384) {
        hipLaunchKernelGGL(finalize_flat_kernel<Map>, dim3((len + 255) / 256), dim3(256), 0, st, part, nblk, len, map);
        return;
    }
    if (len <= 2048 && nblk >= 128) {
        hipLaunchKernelGGL((finalize_kernel<Map, 16>), dim3((len + 15) / 16), dim3(1024), 0, st, part, nblk, len, map);
        return;
    }
    hipLaunchKernelGGL((finalize_kernel<Map, FIN_COLS>), dim3((len + FIN_COLS - 1) / FIN_COLS), dim3(FIN_COLS * FIN_SLICES), 0, st,
                       part, nblk, len, map);
}

template <typename T>
struct MapVec {  // out[j] = v
    T *out;
    __device__ void operator()(int j, double v) const { out[j] = (T)v; }
};
template <typename T>
struct MapSplit2 {  // columns [0,len1) -> out1, rest -> out2
    T *out1, *out2;
    int len1;
    __device__ void operator()(int j, double v) const {
        if (j < len1) out1[j] = (T)v; else out2[j - len1] = (T)v;
    }
};

template <> struct RiderOf<MapVec<float>> {
    static constexpr bool ok = true;
    static PtvRider make(const MapVec<float> &m) { PtvRider r{}; r.kind = RIDER_VEC; r.p[0] = m.out; return r; }
};
template <> struct RiderOf<MapSplit2<float>> {
    static constexpr bool ok = true;
    static PtvRider make(const MapSplit2<float> &m) {
        PtvRider r{}; r.kind = RIDER_SPLIT2; r.p[0] = m.out1; r.p[1] = m.out2; r.i0 = m.len1; return r;
    }
};

// logits backward: partials [nblk][c][G+4] -> gM (c,G), ga (c,3), gb (c)
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
template <> struct RiderOf<MapLogitsParams> {
    static constexpr bool ok = true;
    static PtvRider make(const MapLogitsParams &m) {
        PtvRider r{}; r.kind = RIDER_LOGITS_PARAMS; r.p[0] = m.gM; r.p[1] = m.ga; r.p[2] = m.gb; r.i0 = m.g; return r;
    }
};

// fused logits backward (gva_bwd_logits.hip): columns [0, c (g+4)) as MapLogitsParams, then 16-padded grad cW
struct MapLogitsFused {
    float *gM, *ga, *gb, *gcW;
    int g, c;
    __device__ void operator()(int e, double v) const {
        const int per = g + 4, np = c * per;
        if (e < np) {
            const int ch = e / per, j = e - ch * per;
            if (j < g) gM[ch * g + j] = (float)v;
            else if (j < g + 3) ga[ch * 3 + (j - g)] = (float)v;
            else gb[ch] = (float)v;
        } else if (e - np < g) gcW[e - np] = (float)v;
    }
};
template <> struct RiderOf<MapLogitsFused> {
    static constexpr bool ok = true;
    static PtvRider make(const MapLogitsFused &m) {
        PtvRider r{}; r.kind = RIDER_LOGITS_FUSED; r.p[0] = m.gM; r.p[1] = m.ga; r.p[2] = m.gb; r.p[3] = m.gcW; r.i0 = m.g; r.i1 = m.c;
        return r;
    }
};

// fused softmax / aggregation backward: columns of the workgroup record -> ga (c,3), gb (c), gsc, gsh, gWw2 (g,g), gbw2
struct MapBwdPoint {
    float *ga, *gb, *gsc, *gsh, *gWw2, *gbw2;
    int c, g;
    __device__ void operator()(int e, double v) const {
        if (e < 4 * c) {
            const int ch = e >> 2, j = e & 3;
            if (j < 3) ga[ch * 3 + j] = (float)v; else gb[ch] = (float)v;
        } else if (e < 4 * c + g) gsc[e - 4 * c] = (float)v;
        else if (e < 4 * c + 2 * g) gsh[e - 4 * c - g] = (float)v;
        else if (e < 4 * c + 2 * g + g * g) gWw2[e - 4 * c - 2 * g] = (float)v;
        else gbw2[e - 4 * c - 2 * g - g * g] = (float)v;
    }
};

template <> struct RiderOf<MapBwdPoint> {
    static constexpr bool ok = true;
    static PtvRider make(const MapBwdPoint &m) {
        PtvRider r{}; r.kind = RIDER_BWD_POINT;
        r.p[0] = m.ga; r.p[1] = m.gb; r.p[2] = m.gsc; r.p[3] = m.gsh; r.p[4] = m.gWw2; r.p[5] = m.gbw2; r.i0 = m.c; r.i1 = m.g;
        return r;
    }
};

// up to six weight gradients (+ bias sums) of one shape from one split-K record [count][wlen] weights, [count][cout] biases.
// Named fields and compare chains on purpose: an array member indexed with a runtime value is placed in scratch memory,
// and a host kernel that needs a scratch allocation is dispatched more slowly for ALL of its workgroups (measured:
// +0.15 ms per step with 120 bytes of scratch in the two host kernels).
struct MapWgradN {
    float *w0, *w1, *w2, *w3, *w4, *w5, *b0, *b1, *b2, *b3, *b4, *b5;
    int count, wlen, cout;
    __device__ void operator()(int e, double v) const {
        const int wtot = count * wlen;
        if (e < wtot) {
            const int b = e / wlen;
            float *p = b == 0 ? w0 : b == 1 ? w1 : b == 2 ? w2 : b == 3 ? w3 : b == 4 ? w4 : w5;
            p[e - b * wlen] = (float)v;
        } else {
            const int r = e - wtot, b = r / cout;
            float *p = b == 0 ? b0 : b == 1 ? b1 : b == 2 ? b2 : b == 3 ? b3 : b == 4 ? b4 : b5;
            if (p) p[r - b * cout] = (float)v;
        }
    }
};

// column sums [rb * COLS, ...) of a rider's records by one 256-thread workgroup: one thread per column for few records,
// 16 columns x 16 record slices otherwise (slice sums combined in slice order: fixed association, as finalize_kernel)
template <class Map>
__device__ __forceinline__ void rider_columns(const PtvRider &R, int rb, Map map) {
    const float *__restrict__ part = R.part;
    const int len = R.len, nblk = R.nblk;
    if (nblk <= 32) {
        const int j = rb * 256 + threadIdx.x;
        if (j >= len) return;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int b = 0;
        for (; b + 3 < nblk; b += 4) {
            a0 += (double)part[(size_t)b * len + j];
            a1 += (double)part[(size_t)(b + 1) * len + j];
            a2 += (double)part[(size_t)(b + 2) * len + j];
            a3 += (double)part[(size_t)(b + 3) * len + j];
        }
        for (; b < nblk; ++b) a0 += (double)part[(size_t)b * len + j];
        map(j, (a0 + a1) + (a2 + a3));
        return;
    }
    // 16 columns x 16 record slices, or 8 x 32 from 1 024 records on (the host side sizes R.blocks accordingly)
    __shared__ double s_rider[256];
    const int cols = nblk >= 1024 ? 8 : 16, S = 256 / cols;
    const int col = threadIdx.x & (cols - 1), sl = threadIdx.x / cols;
    const int j = rb * cols + col;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (j < len) {
        int b = sl;
        for (; b + 7 * S < nblk; b += 8 * S) {  // eight loads in flight, the four-chain order (see finalize_kernel)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + S * u) * len + j];
            a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
            a0 += (double)v[4]; a1 += (double)v[5]; a2 += (double)v[6]; a3 += (double)v[7];
        }
        for (; b + 3 * S < nblk; b += 4 * S) {
            a0 += (double)part[(size_t)b * len + j];
            a1 += (double)part[(size_t)(b + S) * len + j];
            a2 += (double)part[(size_t)(b + 2 * S) * len + j];
            a3 += (double)part[(size_t)(b + 3 * S) * len + j];
        }
        for (; b < nblk; b += S) a0 += (double)part[(size_t)b * len + j];
    }
    s_rider[sl * cols + col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && j < len) {
        double v = 0.0;
        for (int t = 0; t < S; ++t) v += s_rider[t * cols + col];
        map(j, v);
    }
}

// workgroup `rb` of the riders' workgroups (a host kernel's blockIdx.x - its own workgroup count); blockDim.x == 256
__device__ __forceinline__ void rider_run(const PtvRiders &Rs, int rb) {
    for (int i = 0; i < Rs.count; ++i) {
        const PtvRider &R = Rs.r[i];
        if (rb < R.blocks) {
            switch (R.kind) {  // uniform over the workgroup
                case RIDER_VEC: rider_columns(R, rb, MapVec<float>{R.p[0]}); break;
                case RIDER_SPLIT2: rider_columns(R, rb, MapSplit2<float>{R.p[0], R.p[1], R.i0}); break;
                case RIDER_LOGITS_PARAMS: rider_columns(R, rb, MapLogitsParams{R.p[0], R.p[1], R.p[2], R.i0}); break;
                case RIDER_LOGITS_FUSED: rider_columns(R, rb, MapLogitsFused{R.p[0], R.p[1], R.p[2], R.p[3], R.i0, R.i1}); break;
                case RIDER_BWD_POINT: rider_columns(R, rb, MapBwdPoint{R.p[0], R.p[1], R.p[2], R.p[3], R.p[4], R.p[5], R.i0, R.i1}); break;
                case RIDER_WGRADN:
                    rider_columns(R, rb, MapWgradN{R.p[0], R.p[1], R.p[2], R.p[3], R.p[4], R.p[5], R.p[6], R.p[7], R.p[8], R.p[9],
                                                   R.p[10], R.p[11], R.i2, R.i0, R.i1});
                    break;
                default: break;
            }
            return;
        }
        rb -= R.blocks;
    }
}
inline int rider_blocks(const PtvRiders &Rs) {
    int b = 0;
    for (int i = 0; i < Rs.count; ++i) b += Rs.r[i].blocks;
    return b;
}

// BatchNorm over the (N*K, G) logits from their column sums T1, T2 (gva_fold.hip: fold_w).  sc == NULL: absent.
// The block runtime passes it to the logits stage, whose final reduction then also emits the folded affine
// (one launch less per attention block than a separate fold kernel).
struct FoldWFwdArgs {
    const float *gamma, *beta;
    float *run_mean, *run_var;
    long long *batches;
    int training;
    double rows;
    float eps, momentum;
    float *sc, *sh;
    double *mean_out, *rstd_out;
};

__device__ inline void fold_w_fwd_channel(const FoldWFwdArgs &A, int j, double t1, double t2) {
    double mean, rstd;
    if (A.training) {
        mean = t1 / A.rows;
        double var = t2 / A.rows - mean * mean;
        var = var > 0.0 ? var : 0.0;
        rstd = 1.0 / sqrt(var + (double)A.eps);
        if (A.run_mean) {
            const double unb = A.rows > 1.0 ? var * (A.rows / (A.rows - 1.0)) : var;
            A.run_mean[j] = (float)((1.0 - A.momentum) * (double)A.run_mean[j] + A.momentum * mean);
            A.run_var[j] = (float)((1.0 - A.momentum) * (double)A.run_var[j] + A.momentum * unb);
            if (j == 0 && A.batches) *A.batches += 1;
        }
    } else {
        mean = (double)A.run_mean[j];
        rstd = 1.0 / sqrt((double)A.run_var[j] + (double)A.eps);
    }
    const double s = (double)A.gamma[j] * rstd;
    A.sc[j] = (float)s;
    A.sh[j] = (float)((double)A.beta[j] - mean * s);
    A.mean_out[j] = mean;
    A.rstd_out[j] = rstd;
}

// Backward of fold_w (the BatchNorm over the logits) for group j: gradients of the column sums T1, T2 that the rows kernel of
// the logits backward folds into every row gradient, and the BatchNorm's own parameter gradients.  gsc == NULL: absent (the
// caller supplies gT1 / gT2 arrays).  The block runtime passes it to the rows kernel, which evaluates it in its prologue
// (g <= 64 values per thread) instead of waiting for a launch of its own (gva_fold.hip: fold_w_bwd_kernel).
struct FoldWBwdArgs {
    const float *gamma;
    const double *mean, *rstd;
    int training;
    double rows;
    const float *gsc, *gsh;
    float *ggamma, *gbeta;
};

__device__ __forceinline__ void fold_w_bwd_channel(const FoldWBwdArgs &A, int j, double &gT1, double &gT2, float &ggamma,
                                                   float &gbeta) {
    const double mean = A.mean[j], rstd = A.rstd[j], gam = A.gamma[j];
    const double gs = (double)A.gsc[j] - (double)A.gsh[j] * mean;  // d/ds of (sc = s, sh = beta - mean s)
    gbeta = A.gsh[j];
    ggamma = (float)(gs * rstd);
    if (A.training) {
        const double gvar = gs * gam * (-0.5) * rstd * rstd * rstd;
        const double gmean = -(double)A.gsh[j] * gam * rstd + gvar * (-2.0 * mean);
        gT1 = gmean / A.rows;
        gT2 = gvar / A.rows;
    } else {
        gT1 = 0.0;
        gT2 = 0.0;
    }
}

// "last block" tail of the logits kernels: column sums -> T1, T2 (-> folded affine)
__device__ __forceinline__ void finalize_logit_sums(const float *part, int nblk, int g, double *T1, double *T2,
                                                    const FoldWFwdArgs &F) {
    finalize_columns(part, nblk, 2 * g, MapSplit2<double>{T1, T2, g});
    if (F.sc) {
        __threadfence_block();
        __syncthreads();
        if ((int)threadIdx.x < g) fold_w_fwd_channel(F, threadIdx.x, T1[threadIdx.x], T2[threadIdx.x]);
    }
}

// the same as a launch of its own (large grids): a workgroup owns 4 groups = 8 of the 2g record columns (T1 and T2 of a
// group are finished by the same workgroup, which then folds that group's BatchNorm), 128 record slices each; slice sums
// are combined in slice order.  One 1024-thread workgroup for all columns walked nblk * 2g / 1024 records per thread:
// 8 - 9 us on the critical path of every attention block.
constexpr int FLS_GROUPS = 4, FLS_SLICES = 1024 / (2 * FLS_GROUPS);
static __global__ __launch_bounds__(1024) void finalize_logit_sums_kernel(const float *__restrict__ part, int nblk, int g, double *T1,
                                                                          double *T2, FoldWFwdArgs F) {
    __shared__ double s_acc[FLS_SLICES][2 * FLS_GROUPS];
    __shared__ double s_sum[2 * FLS_GROUPS];
    const int len = 2 * g;
    const int cl = threadIdx.x % (2 * FLS_GROUPS), sl = threadIdx.x / (2 * FLS_GROUPS);
    const int j = blockIdx.x * FLS_GROUPS + (cl % FLS_GROUPS);  // group
    const int col = cl < FLS_GROUPS ? j : g + j;                // its T1 or T2 column
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (j < g) {
        int b = sl;
        for (; b + 3 * FLS_SLICES < nblk; b += 4 * FLS_SLICES) {
            a0 += (double)part[(size_t)b * len + col];
            a1 += (double)part[(size_t)(b + FLS_SLICES) * len + col];
            a2 += (double)part[(size_t)(b + 2 * FLS_SLICES) * len + col];
            a3 += (double)part[(size_t)(b + 3 * FLS_SLICES) * len + col];
        }
        for (; b < nblk; b += FLS_SLICES) a0 += (double)part[(size_t)b * len + col];
    }
    s_acc[sl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0) {
        double v = 0.0;
#pragma unroll 16
        for (int t = 0; t < FLS_SLICES; ++t) v += s_acc[t][cl];
        s_sum[cl] = v;
        if (j < g) { if (cl < FLS_GROUPS) T1[j] = v; else T2[j] = v; }
    }
    __syncthreads();
    if (F.sc && (int)threadIdx.x < FLS_GROUPS && (int)(blockIdx.x * FLS_GROUPS + threadIdx.x) < g)
        fold_w_fwd_channel(F, blockIdx.x * FLS_GROUPS + threadIdx.x, s_sum[threadIdx.x], s_sum[FLS_GROUPS + threadIdx.x]);
}

}  // namespace gva
