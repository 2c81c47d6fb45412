// ao_amd/csrc/model.hip -- PointTransformerV2.forward / backward (point_transformer_v2m2_base.py:556-576) as ONE native
// call per direction.
//
//   forward:   x = ReLU(BN(feat We^T))                               GVAPatchEmbed.proj (:424-428, :441-444)
//              x = blocks(seq 0)                                     patch_embed.blocks
//              for i in 0..S-1:  d = ReLU(BN(x Wd^T)); x = segment-max of d over the voxels of level i (GridPool :244-269)
//                                x = blocks(seq 1+i);  skip[i+1] = x
//              for i in S-1..0:  u = ReLU(BN(x Wu^T + bu)); s = ReLU(BN(skip[i] Ws^T + bs))
//                                x = interp(u) + s   or   u[cluster] + s      (UnpoolWithSkip :305-316)
//                                x = blocks(seq 1+S+i)
//              logits = ReLU(BN(x Wh^T + bh)) Wc^T + bc                       (seg_head :545-554)
//   backward:  the same chain reversed.  The gradient of a skip tensor has two contributions (the decoder's proj_skip and
//              the encoder's GridPool.fc); the second is added by the row GEMM's accumulate epilogue.
//
// Every kernel is enqueued on the caller's stream from here (block.hip runs the Blocks); nothing synchronises, nothing is
// allocated: activations live in the caller-owned `saved` arena (288 GB of HBM: nothing is recomputed), temporaries in
// the caller's workspace.  Parameter gradients go straight to the destinations named in the ptv2_model (slots of the
// optimizer's flat gradient buffer in ao_amd/ptv2/native_model.py): no per-parameter tensors, no flatten copy.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

int ptv2_blocks_fold_forward(int count, const ptv2_block *blocks, void *stream);  // block.hip
void ptv2_gva_set_prefolded(int on);                                                // gva_block.hip
size_t ptv2_gva_fold_scratch_floats(int c, int g);  // deferred M / cW glue of the attention backward (gva_block.hip)
void ptv2_gva_set_fold_scratch(float *p);
int ptv2_gva_flush_folds(void *stream);
void ptv2_gva_drop_folds();

namespace {

constexpr int TPB = 256;

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

// y[r, o] = sum_i x[r, i] W[o, i] + b[o] for a narrow side (cin or cout not a multiple of 4, or tiny): the patch embedding's
// Linear(in_channels = 6 or 9, c0) and the classifier Linear(c0, num_classes = 13 or 20).  One thread per output element,
// W and b in LDS; the x row of a thread is shared with its cout neighbours through L1.
__global__ __launch_bounds__(TPB) void small_linear_fwd_kernel(long long n, int cin, int cout, const float *__restrict__ x,
                                                               const float *__restrict__ W, const float *__restrict__ b,
                                                               float *__restrict__ y) {
    extern __shared__ float lds[];
    // W rows at an ODD pitch: the lanes of a wavefront hold consecutive outputs o and read w[o * pitch + i] -- with the natural
    // pitch cin = 6 (the S3DIS patch embedding) the 64 lanes fell on 32 banks, two each (SQ_LDS_BANK_CONFLICT 0.47 of this
    // kernel's LDS cycles, profiles/r04_final_sq_counters.jsonl); an odd pitch is co-prime with the 64 banks
    const int pitch = cin | 1;
    float *w = lds, *bias = lds + (size_t)cout * pitch;
    for (int e = threadIdx.x; e < cout * cin; e += TPB) { const int o = e / cin; w[o * pitch + (e - o * cin)] = W[e]; }
    for (int e = threadIdx.x; e < cout; e += TPB) bias[e] = b ? b[e] : 0.f;
    __syncthreads();
    const long long total = n * cout;
    const bool narrow = total < (1LL << 31);  // 32-bit index arithmetic (a 64-bit division per element costs more than the FMAs)
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const long long r = narrow ? (long long)((unsigned)e / (unsigned)cout) : e / cout;
        const int o = (int)(e - r * cout);
        const float *xr = x + r * cin, *wr = w + (size_t)o * pitch;
        float acc = bias[o];
        for (int i = 0; i < cin; ++i) acc = __builtin_fmaf(xr[i], wr[i], acc);
        y[e] = acc;
    }
}

// the classifier (cin % 4 == 0, COUT = 13 / 20 classes): one thread per ROW keeps the COUT sums in registers, reads its
// x row as float4 and the W quads as LDS broadcasts.  One thread per output element (above) re-reads the row COUT times and
// its W reads conflict 6-way (row pitch cin = 48 floats): 31 us at 120 k points against 12 us here
template <int COUT>
__global__ __launch_bounds__(TPB) void classifier_fwd_kernel(long long n, int cin, const float *__restrict__ x,
                                                             const float *__restrict__ W, const float *__restrict__ b,
                                                             float *__restrict__ y) {
    extern __shared__ float4 lds4c[];
    float *w = (float *)lds4c;
    for (int e = threadIdx.x; e < COUT * cin; e += TPB) w[e] = W[e];
    __syncthreads();
    const int cq = cin >> 2;
    for (long long r = (long long)blockIdx.x * TPB + threadIdx.x; r < n; r += (long long)gridDim.x * TPB) {
        const float4 *xr = (const float4 *)(x + r * cin);
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = b ? b[o] : 0.f;
        for (int q = 0; q < cq; ++q) {
            const float4 v = xr[q];
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float4 ww = *(const float4 *)(w + (size_t)o * cin + 4 * q);
                acc[o] = __builtin_fmaf(v.x, ww.x, acc[o]); acc[o] = __builtin_fmaf(v.y, ww.y, acc[o]);
                acc[o] = __builtin_fmaf(v.z, ww.z, acc[o]); acc[o] = __builtin_fmaf(v.w, ww.w, acc[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < COUT; ++o) y[r * COUT + o] = acc[o];
    }
}

// gx[r, i] = sum_o gy[r, o] W[o, i]: the input gradient of the classifier
__global__ __launch_bounds__(TPB) void small_linear_bwd_kernel(long long n, int cin, int cout, const float *__restrict__ gy,
                                                               const float *__restrict__ W, float *__restrict__ gx) {
    extern __shared__ float lds[];
    for (int e = threadIdx.x; e < cout * cin; e += TPB) lds[e] = W[e];
    __syncthreads();
    const long long total = n * cin;
    const bool narrow = total < (1LL << 31);
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const long long r = narrow ? (long long)((unsigned)e / (unsigned)cin) : e / cin;
        const int i = (int)(e - r * cin);
        const float *g = gy + r * cout;
        float acc = 0.f;
        for (int o = 0; o < cout; ++o) acc = __builtin_fmaf(g[o], lds[(size_t)o * cin + i], acc);
        gx[e] = acc;
    }
}

// the same, four consecutive inputs i per thread (cin % 4 == 0): the gy row is read once per quad, W as float4
__global__ __launch_bounds__(TPB) void small_linear_bwd4_kernel(long long n, int cin, int cout, const float *__restrict__ gy,
                                                                const float *__restrict__ W, float4 *__restrict__ gx) {
    extern __shared__ float4 lds4b[];
    float *w = (float *)lds4b;
    for (int e = threadIdx.x; e < cout * cin; e += TPB) w[e] = W[e];
    __syncthreads();
    const int cq = cin >> 2;
    const long long total = n * cq;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const long long r = (long long)((unsigned)e / (unsigned)cq);  // the launcher keeps n * cin / 4 < 2^31
        const int q = (int)(e - r * cq);
        const float *g = gy + r * cout;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int o = 0; o < cout; ++o) {
            const float gv = g[o];
            const float4 ww = *(const float4 *)(w + (size_t)o * cin + 4 * q);
            acc.x = __builtin_fmaf(gv, ww.x, acc.x); acc.y = __builtin_fmaf(gv, ww.y, acc.y);
            acc.z = __builtin_fmaf(gv, ww.z, acc.z); acc.w = __builtin_fmaf(gv, ww.w, acc.w);
        }
        gx[e] = acc;
    }
}

// out[i, :] += u[cluster[i], :]: the "map" unpool onto the skip branch (:309-310, :315)
__global__ __launch_bounds__(TPB) void gather_add_rows_kernel(long long total4, int c4, const float4 *__restrict__ u,
                                                              const long long *__restrict__ cluster, float4 *out) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total4; e += (long long)gridDim.x * TPB) {
        const long long r = e / c4;
        const int q = (int)(e - r * c4);
        const float4 a = u[cluster[r] * c4 + q];
        float4 o = out[e];
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        out[e] = o;
    }
}

__global__ void bn_eval_moments_kernel(int c, const float *__restrict__ rm, const float *__restrict__ rv, float eps,
                                       float *__restrict__ mean, float *__restrict__ rstd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) {
        mean[i] = rm[i];
        rstd[i] = 1.0f / sqrtf(rv[i] + eps);
    }
}

int grid_for(long long total) { return (int)std::min<long long>((total + TPB - 1) / TPB, 256 * 8); }

struct LinBnSaved {
    float *h, *y, *mean, *rstd;  // pre-BatchNorm rows, ReLU(BN(h)), batch (or running) moments
};

struct Arena {  // the `saved` buffer of one forward
    LinBnSaved embed, down[PTV2_MAX_STAGES], up[PTV2_MAX_STAGES], up_skip[PTV2_MAX_STAGES], head;
    float *pooled[PTV2_MAX_STAGES];  // (n_{i+1}, c_{i+1})
    int *arg[PTV2_MAX_STAGES];       // arg-max rows of the pooling
    float *block_y[PTV2_MAX_BLOCKS];
    char *block_saved[PTV2_MAX_BLOCKS];
    size_t block_saved_bytes[PTV2_MAX_BLOCKS];
    size_t bytes, bytes0;  // of `saved`; of `saved0` (0 when the prefix lives at the head of `saved`)
};

bool model_ok(const ptv2_model *M) {
    if (!M || M->num_stages < 1 || M->num_stages > PTV2_MAX_STAGES || M->num_blocks < 1 || M->num_blocks > PTV2_MAX_BLOCKS)
        return false;
    if (M->in_channels < 1 || M->num_classes < 1 || M->in_channels > 64 || M->num_classes > 256) return false;
    const int S = M->num_stages;
    for (int i = 0; i <= S; ++i)
        if (M->level[i].n < 2 || !M->level[i].coord) return false;
    for (int q = 0; q <= 2 * S; ++q) {
        const ptv2_seq &s = M->seq[q];
        if (s.depth < 0 || s.first_block < 0 || s.first_block + s.depth > M->num_blocks || s.level < 0 || s.level > S) return false;
        if (s.depth > 0 && (!s.idx || s.c < 4 || s.g < 1 || s.k < 1)) return false;
    }
    if (M->embed.cin != M->in_channels || (M->seq[0].depth > 0 && M->embed.cout != M->seq[0].c)) return false;
    return M->feat && M->logits && M->head_w;
}

// what the prefix (patch embedding + seq 0) reads of the struct
bool prefix_ok(const ptv2_model *M) {
    if (!M || M->num_blocks < 1 || M->num_blocks > PTV2_MAX_BLOCKS || M->in_channels < 1 || M->in_channels > 64) return false;
    if (M->level[0].n < 2 || !M->level[0].coord) return false;
    const ptv2_seq &s = M->seq[0];
    if (s.depth < 0 || s.first_block != 0 || s.depth > M->num_blocks || s.level != 0) return false;
    if (s.depth > 0 && (!s.idx || s.c < 4 || s.g < 1 || s.k < 1)) return false;
    if (M->embed.cin != M->in_channels || (s.depth > 0 && M->embed.cout != s.c)) return false;
    return M->feat != nullptr && !M->checkpoint;
}

// base0 != NULL (or split): the prefix's items -- the patch embedding's rows, the outputs and saved regions of seq 0's Blocks --
// are carved from a region of their own, whose layout depends on level 0 alone; prefix_only: nothing else is looked at
Arena carve(const ptv2_model *M, void *base, void *base0 = nullptr, bool split = false, bool prefix_only = false) {
    Arena A{};
    split = split || base0 != nullptr;
    char *p = (char *)base, *p0 = (char *)base0;
    size_t off = 0, off0 = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += al(bytes); return r; };
    auto take0 = [&](size_t bytes) {
        if (!split) return take(bytes);
        char *r = p0 ? p0 + off0 : nullptr;
        off0 += al(bytes);
        return r;
    };
    auto linbn = [&](const ptv2_linbn &L, int n, bool keep_y) {
        LinBnSaved s;
        s.h = (float *)take(sizeof(float) * (size_t)n * L.cout);
        s.y = keep_y ? (float *)take(sizeof(float) * (size_t)n * L.cout) : nullptr;
        s.mean = (float *)take(sizeof(float) * L.cout);
        s.rstd = (float *)take(sizeof(float) * L.cout);
        return s;
    };
    const int S = M->num_stages;
    {   // the prefix first
        const ptv2_linbn &L = M->embed;
        const int n = M->level[0].n;
        A.embed.h = (float *)take0(sizeof(float) * (size_t)n * L.cout);
        A.embed.y = (float *)take0(sizeof(float) * (size_t)n * L.cout);
        A.embed.mean = (float *)take0(sizeof(float) * L.cout);
        A.embed.rstd = (float *)take0(sizeof(float) * L.cout);
        const ptv2_seq &s = M->seq[0];
        if (!M->checkpoint)
            for (int j = 0; j < s.depth; ++j) {
                const int b = s.first_block + j;
                A.block_y[b] = (float *)take0(sizeof(float) * (size_t)n * s.c);
                A.block_saved_bytes[b] = ptv2_block_saved_bytes(n, s.k, s.c, s.g);
                A.block_saved[b] = take0(A.block_saved_bytes[b]);
            }
    }
    if (prefix_only) {
        A.bytes = off;
        A.bytes0 = off0;
        return A;
    }
    for (int i = 0; i < S; ++i) {
        A.down[i] = linbn(M->down[i], M->level[i].n, true);
        A.pooled[i] = (float *)take(sizeof(float) * (size_t)M->level[i + 1].n * M->down[i].cout);
        A.arg[i] = (int *)take(sizeof(int) * (size_t)M->level[i + 1].n * M->down[i].cout);
        A.up[i] = linbn(M->up[i], M->level[i + 1].n, true);
        A.up_skip[i] = linbn(M->up_skip[i], M->level[i].n, true);  // y = the unpool output (skip branch + unpooled rows)
    }
    A.head = linbn(M->head, M->level[0].n, true);
    size_t shared_saved = 0;  // checkpointing: one region, sized for the largest Block, shared by all of them
    for (int q = (M->checkpoint ? 0 : 1); q <= 2 * S; ++q) {  // (seq 0: above, unless its Blocks share the checkpoint region)
        const ptv2_seq &s = M->seq[q];
        const int n = M->level[s.level].n;
        for (int j = 0; j < s.depth; ++j) {
            const int b = s.first_block + j;
            A.block_y[b] = (float *)take(sizeof(float) * (size_t)n * s.c);
            A.block_saved_bytes[b] = ptv2_block_saved_bytes(n, s.k, s.c, s.g);
            if (M->checkpoint) shared_saved = std::max(shared_saved, A.block_saved_bytes[b]);
            else A.block_saved[b] = take(A.block_saved_bytes[b]);
        }
    }
    if (M->checkpoint) {
        char *region = take(shared_saved);
        for (int b = 0; b < M->num_blocks; ++b) A.block_saved[b] = region;
    }
    A.bytes = off;
    A.bytes0 = off0;
    return A;
}

struct Work {
    char *block; size_t block_bytes;   // workspace of the Block runtime
    char *dense; size_t dense_bytes;   // BatchNorm / weight-gradient partial records
    float *ga, *gb, *gc;               // gradient temporaries, max over levels of n * widest channel count
    float *gskip[PTV2_MAX_STAGES + 1]; // gradient of the encoder output at level i (two contributions)
    float *fold_scratch[PTV2_MAX_BLOCKS];    // per-Block operands of the attention's parameter glue, run once at the end
    char *wdefer; size_t wdefer_bytes;       // deferred weight gradients (dense.hip): job table, kept operands, chunk records
    size_t bytes;
};

// prefix_only: the workspace of the forward's prefix (patch embedding + seq 0) -- a function of level 0 alone
Work carve_work(const ptv2_model *M, void *base, bool prefix_only = false) {
    Work W{};
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += al(bytes); return r; };
    const int S = M->num_stages;
    W.block_bytes = 0;
    size_t widest = 0;
    W.dense_bytes = 0;
    for (int q = 0; q <= (prefix_only ? 0 : 2 * S); ++q) {
        const ptv2_seq &s = M->seq[q];
        if (s.depth < 1) continue;
        const int n = M->level[s.level].n;
        W.block_bytes = std::max(W.block_bytes, ptv2_block_workspace_bytes(n, s.k, s.c, s.g));
        widest = std::max(widest, (size_t)n * s.c);
    }
    auto note = [&](const ptv2_linbn &L, int n) {
        widest = std::max(widest, (size_t)n * std::max(L.cin, L.cout));
        W.dense_bytes = std::max(W.dense_bytes, dense_workspace_bytes(n, std::max(L.cin, L.cout), std::max(L.cin, L.cout)));
    };
    note(M->embed, M->level[0].n);
    if (prefix_only) {
        W.block = take(W.block_bytes);
        W.dense = take(W.dense_bytes);
        W.bytes = off;
        return W;
    }
    note(M->head, M->level[0].n);
    for (int i = 0; i < S; ++i) {
        note(M->down[i], M->level[i].n);
        note(M->up[i], M->level[i + 1].n);
        note(M->up_skip[i], M->level[i].n);
    }
    W.dense_bytes = std::max(W.dense_bytes, dense_workspace_bytes(M->level[0].n, std::max(M->num_classes, M->head.cout),
                                                                  std::max(M->in_channels, M->head.cout)));
    W.block = take(W.block_bytes);
    W.dense = take(W.dense_bytes);
    W.ga = (float *)take(sizeof(float) * widest);
    W.gb = (float *)take(sizeof(float) * widest);
    W.gc = (float *)take(sizeof(float) * widest);
    for (int i = 0; i <= S; ++i) {
        const int c = i == 0 ? M->embed.cout : M->down[i - 1].cout;
        W.gskip[i] = (float *)take(sizeof(float) * (size_t)M->level[i].n * c);
    }
    for (int q = 0; q <= 2 * S; ++q) {
        const ptv2_seq &s = M->seq[q];
        for (int j = 0; j < s.depth; ++j)
            W.fold_scratch[s.first_block + j] = (float *)take(sizeof(float) * ptv2_gva_fold_scratch_floats(s.c, s.g));
    }
    // per Block: six (n, c) gradient operands + the records of the five-product and of the grouped-projection weight gradient
    W.wdefer_bytes = ptv2_wgrad_defer_table_bytes();
    for (int q = 0; q <= 2 * S; ++q) {
        const ptv2_seq &s = M->seq[q];
        const int n = M->level[s.level].n;
        for (int j = 0; j < s.depth; ++j)
            W.wdefer_bytes += al(sizeof(float) * 6 * (size_t)n * s.c) + al(dense_workspace_bytes(n, 5 * s.c, s.c)) +
                              al(dense_workspace_bytes(n, s.c, s.c)) + 2 * al(sizeof(float) * (size_t)n * s.g) +
                              al(dense_workspace_bytes(n, 2 * s.g, s.c)) + 2048;
    }
    {   // the Linear + BatchNorm layers between the stages: gh (n, cout) + records
        auto layer = [&](const ptv2_linbn &L, int n) {
            W.wdefer_bytes += al(sizeof(float) * (size_t)n * L.cout) + al(dense_workspace_bytes(n, L.cout, L.cin)) + 512;
        };
        layer(M->embed, M->level[0].n);
        layer(M->head, M->level[0].n);
        for (int i = 0; i < S; ++i) {
            layer(M->down[i], M->level[i].n);
            layer(M->up[i], M->level[i + 1].n);
            layer(M->up_skip[i], M->level[i].n);
        }
    }
    W.wdefer = take(W.wdefer_bytes);
    W.bytes = off;
    return W;
}

#define RUN(call)                        \
    do {                                 \
        int rc_ = (call);                \
        if (rc_ != PTV2_OK) return rc_;  \
    } while (0)

std::atomic<int> g_wgrad_defer_mode{-1};  // -1: read AO_AMD_WGRAD_DEFER on first use; 0 off; 1 on
bool wgrad_defer_enabled() {
    int m = g_wgrad_defer_mode.load();
    if (m < 0) {
        const char *e = getenv("AO_AMD_WGRAD_DEFER");
        m = (e && e[0] == '0') ? 0 : 1;
        g_wgrad_defer_mode.store(m);
    }
    return m != 0;
}

}  // namespace
int bn_tiles_apply_relu(int n, int c, float *part, const float *gamma, const float *beta, float *mean, float *rstd, float *running_mean,
                        float *running_var, long long *num_batches_tracked, float eps, float momentum, const float *x, float *y,
                        void *stream);  // dense.hip
namespace {
bool use_batch(const ptv2_model *M, const ptv2_linbn &L) { return M->training || !L.run_mean || !L.run_var; }

// h = x W^T + b (row GEMM, or the narrow kernel when cin is not a multiple of 4); y = ReLU(BN(h)).  `y` may differ from
// the arena slot (the skip branch of the unpool writes the unpool's output buffer).
int linbn_forward(const ptv2_model *M, const ptv2_linbn &L, const LinBnSaved &S, int n, const float *x, float *y, const Work &W,
                  void *stream) {
    hipStream_t st = (hipStream_t)stream;
    // at the pooled levels the GEMM's epilogue leaves the column statistics of h per 64 rows and ONE launch merges them and
    // applies BatchNorm + ReLU (three launches otherwise: statistics, finalize, apply)
    if (use_batch(M, L) && L.cin % 4 == 0 && L.cout % 4 == 0 && (n + 63) / 64 <= 512 &&
        sizeof(float) * bn_tiles_floats(n, L.cout) <= W.dense_bytes) {
        const bool track = M->training && L.run_mean && L.run_var;
        const float *xs[1] = {x}, *ws[1] = {L.w}, *bs[1] = {L.b};
        float *ys[1] = {S.h}, *sts[1] = {(float *)W.dense};
        RUN(rows_gemm_fused_hip_launcher(n, L.cout, L.cin, 1, 0, xs, ws, 0, L.b ? bs : nullptr, ys, 0, nullptr, nullptr, sts, stream));
        if (bn_tiles_apply_relu(n, L.cout, (float *)W.dense, L.gamma, L.beta, S.mean, S.rstd, track ? L.run_mean : nullptr,
                                track ? L.run_var : nullptr, track ? L.batches : nullptr, M->eps, M->momentum, S.h, y, stream))
            return PTV2_OK;
        RUN(bn_tiles_finalize_hip_launcher(n, L.cout, (float *)W.dense, L.gamma, L.beta, S.mean, S.rstd, nullptr, nullptr,
                                           track ? L.run_mean : nullptr, track ? L.run_var : nullptr, track ? L.batches : nullptr,
                                           M->eps, M->momentum, stream));
        RUN(bn_apply_hip_launcher(n, L.cout, S.h, S.mean, S.rstd, L.gamma, L.beta, 1, y, stream));
        return PTV2_OK;
    }
    if (L.cin % 4 == 0 && L.cout % 4 == 0) {
        RUN(rows_gemm_hip_launcher(n, L.cout, L.cin, x, L.w, 0, L.b, S.h, 0, stream));
    } else {
        const size_t lds = sizeof(float) * ((size_t)L.cout * (L.cin | 1) + L.cout);
        hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(grid_for((long long)n * L.cout)), dim3(TPB), lds, st, (long long)n, L.cin,
                           L.cout, x, L.w, L.b, S.h);
    }
    if (use_batch(M, L)) {
        const bool track = M->training && L.run_mean && L.run_var;
        RUN(bn_forward_hip_launcher(n, L.cout, S.h, L.gamma, L.beta, 1, S.mean, S.rstd, track ? L.run_mean : nullptr,
                                    track ? L.run_var : nullptr, track ? L.batches : nullptr, M->eps, M->momentum, nullptr, nullptr,
                                    y, W.dense, W.dense_bytes, stream));
    } else {
        hipLaunchKernelGGL(bn_eval_moments_kernel, dim3(divup(L.cout, 256)), dim3(256), 0, st, L.cout, (const float *)L.run_mean,
                           (const float *)L.run_var, M->eps, S.mean, S.rstd);
        RUN(bn_apply_hip_launcher(n, L.cout, S.h, S.mean, S.rstd, L.gamma, L.beta, 1, y, stream));
    }
    return PTV2_OK;
}

// gy (n,cout) -> BatchNorm + ReLU backward -> gh (tmp); dgamma, dbeta; dW, db from (gh, x); gx (n,cin) = gh W (+)= when
// `accumulate`; gx == NULL: the input needs no gradient (the patch embedding)
int linbn_backward(const ptv2_model *M, const ptv2_linbn &L, const LinBnSaved &S, int n, const float *x, const float *gy, float *gh,
                   float *gx, int accumulate, const Work &W, void *stream) {
    // (weight gradients deferred, dense.hip: gh, the operand of this layer's, lives in the deferral arena until the backward ends)
    float *kept = ptv2_wgrad_defer_active() ? ptv2_wgrad_defer_alloc((size_t)n * L.cout) : nullptr;
    if (kept) gh = kept;
    RUN(bn_backward_hip_launcher(n, L.cout, S.h, gy, S.mean, S.rstd, L.gamma, L.beta, 1, use_batch(M, L) ? 1 : 0, gh, L.ggamma,
                                 L.gbeta, W.dense, W.dense_bytes, stream));
    if (gx) {
        if (L.cin % 4 != 0 || L.cout % 4 != 0) return PTV2_ERR_ARG;
        RUN(rows_gemm_hip_launcher(n, L.cin, L.cout, gh, L.w, 1, nullptr, gx, accumulate, stream));
    }
    ptv2_wgrad_defer_arm(kept != nullptr);
    const int rc = linear_wgrad_hip_launcher(n, L.cout, L.cin, gh, x, L.gw, L.b ? L.gb : nullptr, W.dense, W.dense_bytes, stream);
    ptv2_wgrad_defer_arm(false);
    return rc;
}

void fill_block(const ptv2_model *M, int q, int j, const Arena &A, const float *x, ptv2_block *B) {
    const ptv2_seq &s = M->seq[q];
    const ptv2_level &lv = M->level[s.level];
    const int b = s.first_block + j;
    const ptv2_model_block &mb = M->block[b];
    B->n = lv.n; B->k = s.k; B->c = s.c; B->g = s.g; B->training = M->training;
    B->eps = M->eps; B->momentum = M->momentum;
    B->x = x; B->coord = lv.coord; B->idx = s.idx; B->mu = s.mu; B->cov = s.cov; B->rowscale = M->training ? mb.rowscale : nullptr;
    for (int i = 0; i < PTV2_BLK_NPARAM; ++i) B->param[i] = mb.param[i];
    for (int i = 0; i < PTV2_BLK_NBN; ++i) {
        B->run_mean[i] = mb.run_mean[i]; B->run_var[i] = mb.run_var[i]; B->batches[i] = mb.batches[i];
    }
    B->y = A.block_y[b]; B->saved = A.block_saved[b]; B->saved_bytes = A.block_saved_bytes[b];
    B->matmul_bf16 = M->matmul_bf16;
    B->attn_drop_p = M->training ? mb.attn_drop_p : 0.f; B->attn_drop_seed = mb.attn_drop_seed;
}

// input of block j of sequence q = output of block j-1, or the sequence's input
const float *seq_forward(const ptv2_model *M, int q, const Arena &A, const float *x, const Work &W, void *stream, int *rc) {
    const ptv2_seq &s = M->seq[q];
    for (int j = 0; j < s.depth; ++j) {
        ptv2_block B;
        fill_block(M, q, j, A, x, &B);
        *rc = ptv2_block_forward_hip_launcher(&B, W.block, W.block_bytes, stream);
        if (*rc != PTV2_OK) return nullptr;
        x = B.y;
    }
    *rc = PTV2_OK;
    return x;
}

// gy: gradient of the sequence's output, in one of the ping-pong buffers; returns the buffer holding the gradient of the
// sequence's input (gx_first, when given, receives it directly: the last hop writes there)
float *seq_backward(const ptv2_model *M, int q, const Arena &A, const float *x_in, float *gy, float *other, const Work &W,
                    void *stream, int *rc) {
    const ptv2_seq &s = M->seq[q];
    for (int j = s.depth - 1; j >= 0; --j) {
        ptv2_block B;
        const float *x = j == 0 ? x_in : A.block_y[s.first_block + j - 1];
        fill_block(M, q, j, A, x, &B);
        if (M->checkpoint) {
            // re-run this Block's forward into the shared region (same kernels, same inputs -> the same activations bit for
            // bit).  The reference checkpoints `self.attn` only (point_transformer_v2m2_base.py:169-171), and torch's
            // recomputation runs its BatchNorm layers in training mode again: the running statistics of the four norms INSIDE
            // the attention (q, k, positional bias, weight encoding: 1..4) take the momentum step a second time and their
            // batch counters advance by two per step; norm1 / norm2 / norm3 (0, 5, 6) are outside and move once.
            ptv2_block R = B;
            for (int i = 0; i < PTV2_BLK_NBN; ++i)
                if (M->training && !(i >= 1 && i <= 4)) { R.run_mean[i] = nullptr; R.run_var[i] = nullptr; R.batches[i] = nullptr; }
            *rc = ptv2_block_forward_hip_launcher(&R, W.block, W.block_bytes, stream);
            if (*rc != PTV2_OK) return nullptr;
        }
        ptv2_block_grads G{};
        G.gy = gy; G.inv_ptr = s.inv_ptr; G.inv_rows = s.inv_rows; G.gx = other; G.gparam = nullptr;
        for (int i = 0; i < PTV2_BLK_NPARAM; ++i) G.gp[i] = M->block[s.first_block + j].gparam[i];
        // the attention's parameter-sized glue is queued, not launched (its operands stay in this Block's scratch); not under
        // checkpointing, where the glue reads a saved vector that the next recomputation overwrites
        ptv2_gva_set_fold_scratch(M->checkpoint ? nullptr : W.fold_scratch[s.first_block + j]);
        *rc = ptv2_block_backward_hip_launcher(&B, &G, W.block, W.block_bytes, stream);
        ptv2_gva_set_fold_scratch(nullptr);
        if (*rc != PTV2_OK) return nullptr;
        std::swap(gy, other);
    }
    *rc = PTV2_OK;
    return gy;
}

}  // namespace

extern "C" size_t ptv2_model_saved_bytes(const ptv2_model *M) {
    if (!model_ok(M) || (M->saved0 && M->checkpoint)) return 0;
    return carve(M, nullptr, nullptr, M->saved0 != nullptr).bytes + 256;  // (saved0 set: the rest alone)
}

extern "C" size_t ptv2_model_prefix_saved_bytes(const ptv2_model *M) {
    if (!prefix_ok(M)) return 0;
    return carve(M, nullptr, nullptr, true, true).bytes0 + 256;
}

extern "C" size_t ptv2_model_prefix_workspace_bytes(const ptv2_model *M) {
    if (!prefix_ok(M)) return 0;
    return carve_work(M, nullptr, true).bytes + 256;
}

extern "C" size_t ptv2_model_workspace_bytes(const ptv2_model *M) {
    if (!model_ok(M)) return 0;
    return carve_work(M, nullptr).bytes + 256;
}

namespace {
enum { PHASE_ALL = 0, PHASE_PREFIX = 1, PHASE_REST = 2 };
int model_forward(const ptv2_model *M, void *workspace, size_t workspace_bytes, void *stream, int phase);
int model_backward(const ptv2_model *M, const float *g_logits, void *workspace, size_t workspace_bytes, void *stream);
}  // namespace

// Both directions enqueue their whole kernel sequence as ONE hipGraph launch (graph.hip): the body below runs under stream
// capture, the executable graph of the previous call is updated in place with this call's arguments, and launched.
extern "C" int ptv2_model_forward_hip_launcher(const ptv2_model *M, void *workspace, size_t workspace_bytes, void *stream) {
    PtvGraphScope scope(stream, GRAPH_MODEL_FWD);
    return scope.finish(model_forward(M, workspace, workspace_bytes, scope.stream(), PHASE_ALL));
}

// the same forward as two graphs: what needs level 0 only, and the rest (include/ptv2_hip.h)
extern "C" int ptv2_model_forward_prefix_hip_launcher(const ptv2_model *M, void *workspace, size_t workspace_bytes, void *stream) {
    PtvGraphScope scope(stream, GRAPH_MODEL_FWD_PREFIX);
    return scope.finish(model_forward(M, workspace, workspace_bytes, scope.stream(), PHASE_PREFIX));
}

extern "C" int ptv2_model_forward_rest_hip_launcher(const ptv2_model *M, void *workspace, size_t workspace_bytes, void *stream) {
    PtvGraphScope scope(stream, GRAPH_MODEL_FWD_REST);
    return scope.finish(model_forward(M, workspace, workspace_bytes, scope.stream(), PHASE_REST));
}

extern "C" int ptv2_model_backward_hip_launcher(const ptv2_model *M, const float *g_logits, void *workspace,
                                                size_t workspace_bytes, void *stream) {
    // (with an event that another stream waits on in the body: issued eagerly)
    PtvGraphScope scope(stream, GRAPH_MODEL_BWD, M && !M->decoder_done_event);
    return scope.finish(model_backward(M, g_logits, workspace, workspace_bytes, scope.stream()));
}

namespace {

int model_forward(const ptv2_model *M, void *workspace, size_t workspace_bytes, void *stream, int phase) {
    const bool pre = phase == PHASE_PREFIX;
    if (pre ? (!prefix_ok(M) || !M->saved0) : (!model_ok(M) || !M->saved)) return PTV2_ERR_ARG;
    if (phase == PHASE_REST && (!M->saved0 || M->checkpoint)) return PTV2_ERR_ARG;
    const PtvMatmulScope precision(M->matmul_bf16);
    const Arena A = carve(M, M->saved, M->saved0, pre, pre);
    if ((!pre && M->saved_bytes < A.bytes) || (M->saved0 && M->saved0_bytes < A.bytes0)) return PTV2_ERR_WORKSPACE;
    const Work W = carve_work(M, workspace, pre);
    if (!workspace || workspace_bytes < W.bytes) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int S = M->num_stages;
    int rc = PTV2_OK;
    // the parameter-only folds of every attention block, all at once (they were one 5 us launch on the critical path of
    // each Block)
    // (not under checkpointing: the folds live in the Blocks' saved regions, which then are one shared region -- every Block
    // folds for itself, in its forward and again in its recomputation)
    // (prefix / rest: each folds for the Blocks it runs)
    if (!M->checkpoint) {
        std::vector<ptv2_block> blocks;
        const int q0 = phase == PHASE_REST ? 1 : 0, q1 = pre ? 0 : 2 * S;
        for (int q = q0; q <= q1; ++q)
            for (int j = 0; j < M->seq[q].depth; ++j) {
                ptv2_block B;
                fill_block(M, q, j, A, M->feat /* unused by the folds */, &B);
                blocks.push_back(B);
            }
        RUN(ptv2_blocks_fold_forward((int)blocks.size(), blocks.data(), stream));
    }
    struct Prefolded {
        explicit Prefolded(int on) { ptv2_gva_set_prefolded(on); }
        ~Prefolded() { ptv2_gva_set_prefolded(0); }
    } prefolded(M->checkpoint ? 0 : 1);
    const float *x = nullptr;
    if (phase != PHASE_REST) {
        RUN(linbn_forward(M, M->embed, A.embed, M->level[0].n, M->feat, A.embed.y, W, stream));
        x = seq_forward(M, 0, A, A.embed.y, W, stream, &rc);
        if (rc != PTV2_OK) return rc;
        if (pre) {
            PTV2_CHECK_LAUNCH();
            return PTV2_OK;
        }
    } else {  // the prefix's output, where seq_forward left it
        const ptv2_seq &s0 = M->seq[0];
        x = s0.depth > 0 ? A.block_y[s0.first_block + s0.depth - 1] : A.embed.y;
    }
    const float *skip[PTV2_MAX_STAGES + 1];
    skip[0] = x;
    for (int i = 0; i < S; ++i) {
        const ptv2_level &lv = M->level[i];
        if (!lv.order || !lv.idx_ptr) return PTV2_ERR_ARG;
        RUN(linbn_forward(M, M->down[i], A.down[i], lv.n, x, A.down[i].y, W, stream));
        RUN(pool_max_forward_hip_launcher(M->level[i + 1].n, M->down[i].cout, A.down[i].y, lv.order, lv.idx_ptr, A.pooled[i],
                                          A.arg[i], stream));
        x = seq_forward(M, 1 + i, A, A.pooled[i], W, stream, &rc);
        if (rc != PTV2_OK) return rc;
        skip[i + 1] = x;
    }
    for (int i = S - 1; i >= 0; --i) {
        const ptv2_level &lv = M->level[i];
        const int c = M->up[i].cout;
        if (M->up_skip[i].cout != c || c % 4 != 0) return PTV2_ERR_ARG;
        RUN(linbn_forward(M, M->up[i], A.up[i], M->level[i + 1].n, x, A.up[i].y, W, stream));
        float *out = A.up_skip[i].y;  // skip branch first, the unpooled rows are added onto it
        RUN(linbn_forward(M, M->up_skip[i], A.up_skip[i], lv.n, skip[i], out, W, stream));
        if (M->interp) {
            if (!lv.up_idx || !lv.up_w) return PTV2_ERR_ARG;
            RUN(interpolation_forward_hip_launcher(lv.n, c, 3, A.up[i].y, lv.up_idx, lv.up_w, out, stream));
        } else {
            if (!lv.cluster) return PTV2_ERR_ARG;
            const long long total4 = (long long)lv.n * (c / 4);
            hipLaunchKernelGGL(gather_add_rows_kernel, dim3(grid_for(total4)), dim3(TPB), 0, st, total4, c / 4,
                               (const float4 *)A.up[i].y, lv.cluster, (float4 *)out);
        }
        x = seq_forward(M, 1 + S + i, A, out, W, stream, &rc);
        if (rc != PTV2_OK) return rc;
    }
    RUN(linbn_forward(M, M->head, A.head, M->level[0].n, x, A.head.y, W, stream));
    {
        const int c0 = M->head.cout, nc = M->num_classes;
        const size_t lds = sizeof(float) * ((size_t)nc * (c0 | 1) + nc);  // (covers small_linear_fwd's odd row pitch)
        const int gr = grid_for(M->level[0].n);
        if (c0 % 4 == 0 && nc == 13)
            hipLaunchKernelGGL(classifier_fwd_kernel<13>, dim3(gr), dim3(TPB), lds, st, (long long)M->level[0].n, c0,
                               (const float *)A.head.y, M->head_w, M->head_b, M->logits);
        else if (c0 % 4 == 0 && nc == 20)
            hipLaunchKernelGGL(classifier_fwd_kernel<20>, dim3(gr), dim3(TPB), lds, st, (long long)M->level[0].n, c0,
                               (const float *)A.head.y, M->head_w, M->head_b, M->logits);
        else
            hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(grid_for((long long)M->level[0].n * nc)), dim3(TPB), lds, st,
                               (long long)M->level[0].n, c0, nc, (const float *)A.head.y, M->head_w, M->head_b, M->logits);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

int model_backward(const ptv2_model *M, const float *g_logits, void *workspace, size_t workspace_bytes, void *stream) {
    if (!model_ok(M) || !M->saved || !g_logits || !M->g_head_w) return PTV2_ERR_ARG;
    const PtvMatmulScope precision(M->matmul_bf16);
    if (M->saved0 && M->checkpoint) return PTV2_ERR_ARG;
    const Arena A = carve(M, M->saved, M->saved0);
    if (M->saved_bytes < A.bytes || (M->saved0 && M->saved0_bytes < A.bytes0)) return PTV2_ERR_WORKSPACE;
    const Work W = carve_work(M, workspace);
    if (!workspace || workspace_bytes < W.bytes) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int S = M->num_stages, n0 = M->level[0].n;
    int rc = PTV2_OK;
    ptv2_gva_drop_folds();  // (a previous call that failed half-way may have left entries queued)
    struct DropFolds { bool armed = true; ~DropFolds() { if (armed) ptv2_gva_drop_folds(); } } drop_folds;
    // the Blocks' five-product weight gradients are filed and run by one launch at the end (dense.hip: WgradJob).  Not under
    // checkpointing (their X operands live in the one shared saved region); AO_AMD_WGRAD_DEFER=0: every launch where it is called
    struct WgradDeferScope {
        WgradDeferScope(void *arena, size_t bytes, bool on) { if (on) ptv2_wgrad_defer_begin(arena, bytes); }
        ~WgradDeferScope() { ptv2_wgrad_defer_end(); }
    } wgrad_defer(W.wdefer, W.wdefer_bytes, !M->checkpoint && wgrad_defer_enabled());
    // the sequences' inputs and outputs as the forward wired them
    auto seq_out = [&](int q, const float *in) {
        const ptv2_seq &s = M->seq[q];
        return s.depth > 0 ? (const float *)A.block_y[s.first_block + s.depth - 1] : in;
    };
    const float *skip[PTV2_MAX_STAGES + 1];
    skip[0] = seq_out(0, A.embed.y);
    for (int i = 0; i < S; ++i) skip[i + 1] = seq_out(1 + i, A.pooled[i]);
    const float *dec_in[PTV2_MAX_STAGES + 1];  // input of UnpoolWithSkip i = output of decoder stage i+1 (or the deepest skip)
    dec_in[S] = skip[S];
    for (int i = S - 1; i >= 0; --i) dec_in[i] = seq_out(1 + S + i, A.up_skip[i].y);  // dec_in[i] = output of decoder stage i

    // Gradient buffers: ga / gb ping-pong along the chain, gc holds the gradient in front of a BatchNorm (the operand of
    // the weight gradient), gskip[i] the gradient of the encoder output of level i.
    float *const ga = W.ga, *const gb = W.gb, *const gc = W.gc;
    // head: classifier, then Linear + BatchNorm + ReLU -> gb = gradient of decoder stage 0's output
    {
        const int c0 = M->head.cout, nc = M->num_classes;
        if (c0 % 4 == 0 && (long long)n0 * (c0 / 4) < (1LL << 31))
            hipLaunchKernelGGL(small_linear_bwd4_kernel, dim3(grid_for((long long)n0 * (c0 / 4))), dim3(TPB),
                               sizeof(float) * (size_t)nc * c0, st, (long long)n0, c0, nc, g_logits, M->head_w, (float4 *)ga);
        else
            hipLaunchKernelGGL(small_linear_bwd_kernel, dim3(grid_for((long long)n0 * c0)), dim3(TPB), sizeof(float) * (size_t)nc * c0,
                               st, (long long)n0, c0, nc, g_logits, M->head_w, ga);
        RUN(linear_wgrad_hip_launcher(n0, nc, c0, g_logits, A.head.y, M->g_head_w, M->head_b ? M->g_head_b : nullptr, W.dense, W.dense_bytes, stream));
        RUN(linbn_backward(M, M->head, A.head, n0, dec_in[0], ga, gc, gb, 0, W, stream));
    }
    // decoder stages 0 .. S-1 (the forward ran them S-1 .. 0)
    float *g = gb, *o = ga;
    for (int i = 0; i < S; ++i) {
        const ptv2_level &lv = M->level[i];
        const int c = M->up[i].cout, n_coarse = M->level[i + 1].n;
        float *gin = seq_backward(M, 1 + S + i, A, A.up_skip[i].y, g, o, W, stream, &rc);  // gradient of the unpool's output
        if (rc != PTV2_OK) return rc;
        float *spare = gin == g ? o : g;
        // skip branch: first contribution to the gradient of skip[i]
        RUN(linbn_backward(M, M->up_skip[i], A.up_skip[i], lv.n, skip[i], gin, gc, W.gskip[i], 0, W, stream));
        // unpooled rows: back to the coarse level
        if (M->interp) {
            if (!lv.up_inv_ptr || !lv.up_inv_rows) return PTV2_ERR_ARG;
            RUN(interpolation_backward_gather_hip_launcher(n_coarse, c, 3, gin, lv.up_inv_ptr, lv.up_inv_rows, lv.up_w, spare, stream));
        } else {
            RUN(segment_sum_hip_launcher(n_coarse, c, gin, lv.order, lv.idx_ptr, spare, stream));
        }
        // proj: gradient of the unpool's input = output of decoder stage i+1, or (i+1 == S) the deepest encoder output;
        // gin's rows are dead once both branches have read them
        float *dst = i + 1 == S ? W.gskip[S] : gin;
        RUN(linbn_backward(M, M->up[i], A.up[i], n_coarse, dec_in[i + 1], spare, gc, dst, 0, W, stream));
        g = dst;
        o = spare;
    }
    // head + decoder parameter gradients are final from here on (their finalizes were enqueued above)
    if (M->decoder_done_event) {
        RUN(ptv2_wgrad_defer_flush(stream));
        RUN(ptv2_gva_flush_folds(stream));  // (the decoder Blocks' queued parameter glue belongs to that half)
        if (hipEventRecord((hipEvent_t)M->decoder_done_event, st) != hipSuccess) return PTV2_ERR_LAUNCH;
    }
    // encoder stages S-1 .. 0: gskip[i+1] is complete (skip branch of the decoder + the pooling of stage i+1)
    for (int i = S - 1; i >= 0; --i) {
        const ptv2_level &lv = M->level[i];
        const int c = M->down[i].cout;
        float *gp = seq_backward(M, 1 + i, A, A.pooled[i], W.gskip[i + 1], ga, W, stream, &rc);  // ping-pong gskip[i+1] / ga
        if (rc != PTV2_OK) return rc;
        // pooling: the gradient of a pooled value goes to its arg-max row (all other rows zero)
        (void)ptv2_zero_async(gb, sizeof(float) * (size_t)lv.n * c, st);
        RUN(pool_max_backward_hip_launcher(M->level[i + 1].n, c, gp, A.arg[i], gb, stream));
        // GridPool.fc: the second contribution to the gradient of skip[i], added by the GEMM's accumulate epilogue
        RUN(linbn_backward(M, M->down[i], A.down[i], lv.n, skip[i], gb, gc, W.gskip[i], 1, W, stream));
    }
    // patch embedding (its input needs no gradient)
    {
        float *gp = seq_backward(M, 0, A, A.embed.y, W.gskip[0], ga, W, stream, &rc);
        if (rc != PTV2_OK) return rc;
        RUN(linbn_backward(M, M->embed, A.embed, n0, M->feat, gp, gc, nullptr, 0, W, stream));
    }
    RUN(ptv2_wgrad_defer_flush(stream));  // the Blocks' filed weight gradients: table writers, one launch, one finalize
    RUN(ptv2_gva_flush_folds(stream));  // the queued parameter glue of all attention blocks: two launches
    drop_folds.armed = false;
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

}  // namespace

// 1 / 0: file the Blocks' weight gradients and run them by one launch at the end of the backward / launch each where it is
// called (the A/B switch of tests/test_gpu_native_model.py; also AO_AMD_WGRAD_DEFER=0).  Returns the previous setting; -1 asks.
extern "C" int ptv2_wgrad_defer_mode(int on) {
    const int prev = wgrad_defer_enabled() ? 1 : 0;
    if (on >= 0) g_wgrad_defer_mode.store(on ? 1 : 0);
    return prev;
}
