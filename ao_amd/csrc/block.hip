// ao_amd/csrc/block.hip -- a whole PT-v2m2 Block (point_transformer_v2m2_base.py:131-177) per native call.
//
//   forward:   h1 = x fc1^T            f1 = ReLU(BN1(h1))
//              q = ReLU(BNq(f1 Wq^T+bq))  k = ReLU(BNk(f1 Wk^T+bk))  v = f1 Wv^T+bv
//              attn = GVA(q,k,v,coord,idx)                       (gva_block.hip)
//              f2 = ReLU(BN2(attn))    h3 = f2 fc3^T
//              y  = ReLU(x + rowscale * BN3(h3))
//   BatchNorm is fused into the Linears around it: the GEMM that produces h1 / hq / hk / h3 leaves per-row-block
//   column statistics in its epilogue (one tiny finalize each), and f1, q, k, f2 are never materialised -- their
//   consumers (GEMM, weight gradient, the G-wide projections) apply ReLU(x * sc + sh) on the operand load.
//   backward:  the same chain reversed; sums of several products (g_f1 from q/k/v, g_x from fc1 and the
//              residual) use the GEMM's accumulate epilogue instead of separate adds.
// Everything is enqueued on the caller's stream; the host never synchronises.  The activations the backward
// needs live in one caller-owned `saved` buffer (288 GB of HBM: nothing is recomputed), temporaries in the
// caller's grow-only workspace.
#include <algorithm>

#include "common.h"

size_t bn_tiles_floats_rb(int n, int c, int rb);  // dense.hip: statistics records of rb rows each
void ptv2_skinny_bn_arm(int n, int c, const float *const *x, const float *const *gy, const float *const *mean, const float *const *rstd,
                        const float *const *gamma, const float *const *beta, int relu, void *workspace, size_t workspace_bytes);
void ptv2_skinny_bn_disarm(void);
int gva_block_keeps_A(int k, int c, int g);        // gva_block.hip

namespace {

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

struct Saved {
    float *h1, *hq, *hk, *v, *attn, *h3;                          // (n,c) each
    float *mean[PTV2_BLK_NBN], *rstd[PTV2_BLK_NBN];                // (c): 0,1,2,5,6 used here
    float *bsc[PTV2_BLK_NBN], *bsh[PTV2_BLK_NBN];                  // (c): folded affine of BatchNorm 0,1,2,5
    // GroupedVectorAttention (ptv2_gva_block "saved" fields)
    float *a, *b, *rstd_p, *M, *cW, *kW, *qW, *W1, *w, *A, *sw, *sc, *sh;
    double *mean_w, *rstd_w;
    size_t bytes;
};

Saved carve_saved(void *base, int n, int k, int c, int g) {
    Saved s;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += al(bytes); return r; };
    const size_t nc = sizeof(float) * (size_t)n * c, ng = sizeof(float) * (size_t)n * g;
    const size_t rows = sizeof(float) * (size_t)n * k * g;
    s.h1 = (float *)take(nc); s.hq = (float *)take(nc); s.hk = (float *)take(nc); s.v = (float *)take(nc);
    s.attn = (float *)take(nc); s.h3 = (float *)take(nc);
    for (int i = 0; i < PTV2_BLK_NBN; ++i) {
        s.mean[i] = (float *)take(sizeof(float) * c);
        s.rstd[i] = (float *)take(sizeof(float) * c);
        s.bsc[i] = (float *)take(sizeof(float) * c);
        s.bsh[i] = (float *)take(sizeof(float) * c);
    }
    s.a = (float *)take(sizeof(float) * 3 * c); s.b = (float *)take(sizeof(float) * c);
    s.rstd_p = (float *)take(sizeof(float) * c); s.M = (float *)take(sizeof(float) * (size_t)c * g);
    s.cW = (float *)take(sizeof(float) * g); s.kW = (float *)take(ng); s.qW = (float *)take(ng);
    // (A (n,g,c) only where the attention forward of this shape writes it: not at the deep levels' tile path)
    s.W1 = (float *)take(rows); s.w = (float *)take(rows);
    s.A = gva_block_keeps_A(k, c, g) ? (float *)take(sizeof(float) * (size_t)n * g * c) : nullptr;
    s.sw = (float *)take(ng); s.sc = (float *)take(sizeof(float) * g); s.sh = (float *)take(sizeof(float) * g);
    s.mean_w = (double *)take(sizeof(double) * g); s.rstd_w = (double *)take(sizeof(double) * g);
    s.bytes = off;
    return s;
}

struct Work {
    char *dense; size_t dense_bytes;
    char *gva; size_t gva_bytes;
    float *t[7];  // (n,c) gradient temporaries: the five that feed the batched weight gradient stay alive to the end
    float *stat[4];  // GEMM-epilogue statistics records of h1, hq, hk, h3: ceil(n / 64) x [2][c]
    size_t bytes;
};

Work carve_work(void *base, int n, int k, int c, int g) {
    Work w;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += al(bytes); return r; };
    w.dense_bytes = dense_workspace_bytes(n, 5 * c, c);  // five weight gradients in one launch
    w.dense = take(w.dense_bytes);
    w.gva_bytes = gva_block_workspace_bytes(n, k, c, g);
    w.gva = take(w.gva_bytes);
    for (int i = 0; i < 7; ++i) w.t[i] = (float *)take(sizeof(float) * (size_t)n * c);
    // (records of 16 rows: the deep levels' k-split GEMM and attention tile kernels leave them per 16-row block)
    for (int i = 0; i < 4; ++i) w.stat[i] = (float *)take(sizeof(float) * bn_tiles_floats_rb(n, c, 16));
    w.bytes = off;
    return w;
}

// eval-mode statistics: mean = running_mean, rstd = 1/sqrt(running_var + eps), and the folded affine
__global__ void bn_eval_stats_kernel(int c, const float *__restrict__ rm, const float *__restrict__ rv, float eps,
                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                     float *__restrict__ mean, float *__restrict__ rstd, float *__restrict__ sc,
                                     float *__restrict__ sh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) {
        const float m = rm[i], r = 1.0f / sqrtf(rv[i] + eps);
        mean[i] = m;
        rstd[i] = r;
        sc[i] = r * gamma[i];
        sh[i] = beta[i] - m * r * gamma[i];
    }
}

void param_sizes(int c, int g, long long *sz) {
    const long long cc = (long long)c * c;
    sz[PTV2_BLK_FC1_W] = cc; sz[PTV2_BLK_N1_G] = c; sz[PTV2_BLK_N1_B] = c;
    sz[PTV2_BLK_Q_W] = cc; sz[PTV2_BLK_Q_B] = c; sz[PTV2_BLK_QN_G] = c; sz[PTV2_BLK_QN_B] = c;
    sz[PTV2_BLK_K_W] = cc; sz[PTV2_BLK_K_B] = c; sz[PTV2_BLK_KN_G] = c; sz[PTV2_BLK_KN_B] = c;
    sz[PTV2_BLK_V_W] = cc; sz[PTV2_BLK_V_B] = c;
    sz[PTV2_BLK_P1_W] = 3LL * c; sz[PTV2_BLK_P1_B] = c; sz[PTV2_BLK_PN_G] = c; sz[PTV2_BLK_PN_B] = c;
    sz[PTV2_BLK_P2_W] = cc; sz[PTV2_BLK_P2_B] = c;
    sz[PTV2_BLK_W1_W] = (long long)g * c; sz[PTV2_BLK_W1_B] = g; sz[PTV2_BLK_WN_G] = g; sz[PTV2_BLK_WN_B] = g;
    sz[PTV2_BLK_W2_W] = (long long)g * g; sz[PTV2_BLK_W2_B] = g;
    sz[PTV2_BLK_N2_G] = c; sz[PTV2_BLK_N2_B] = c; sz[PTV2_BLK_FC3_W] = cc; sz[PTV2_BLK_N3_G] = c; sz[PTV2_BLK_N3_B] = c;
}

bool args_ok(const ptv2_block *B) {
    if (!B) return false;
    if (B->n < 2 || B->k < 1 || B->c < 4 || B->c % 4 != 0 || B->c > 1024 || B->g < 1 || B->c % B->g != 0) return false;
    if (!B->x || !B->coord || !B->idx || !B->y || !B->saved) return false;
    if (B->training && (!B->mu || !B->cov)) return false;
    for (int i = 0; i < PTV2_BLK_NPARAM; ++i) {
        const bool optional = i == PTV2_BLK_Q_B || i == PTV2_BLK_K_B || i == PTV2_BLK_V_B;
        if (!B->param[i] && !optional) return false;
    }
    return true;
}

void fill_gva(const ptv2_block *B, const Saved &S, ptv2_gva_block *V) {
    V->n = B->n; V->k = B->k; V->c = B->c; V->g = B->g; V->training = B->training;
    V->eps_p = V->eps_w = B->eps; V->momentum_p = V->momentum_w = B->momentum;
    V->q = S.hq; V->key = S.hk; V->v = S.v;  // pre-BatchNorm projections + the folded affine (applied on operand load)
    V->q_sc = S.bsc[1]; V->q_sh = S.bsh[1]; V->k_sc = S.bsc[2]; V->k_sh = S.bsh[2]; V->coord = B->coord; V->idx = B->idx; V->mu = B->mu; V->cov = B->cov;
    V->Wp1 = B->param[PTV2_BLK_P1_W]; V->bp1 = B->param[PTV2_BLK_P1_B]; V->gamma_p = B->param[PTV2_BLK_PN_G];
    V->beta_p = B->param[PTV2_BLK_PN_B]; V->Wp2 = B->param[PTV2_BLK_P2_W]; V->bp2 = B->param[PTV2_BLK_P2_B];
    V->Ww1 = B->param[PTV2_BLK_W1_W]; V->bw1 = B->param[PTV2_BLK_W1_B]; V->gamma_w = B->param[PTV2_BLK_WN_G];
    V->beta_w = B->param[PTV2_BLK_WN_B]; V->Ww2 = B->param[PTV2_BLK_W2_W]; V->bw2 = B->param[PTV2_BLK_W2_B];
    V->run_mean_p = B->run_mean[3]; V->run_var_p = B->run_var[3]; V->batches_p = B->batches[3];
    V->run_mean_w = B->run_mean[4]; V->run_var_w = B->run_var[4]; V->batches_w = B->batches[4];
    V->out = S.attn;
    V->a = S.a; V->b = S.b; V->rstd_p = S.rstd_p; V->M = S.M; V->cW = S.cW; V->kW = S.kW; V->qW = S.qW; V->W1 = S.W1;
    V->w = S.w; V->A = S.A; V->sw = S.sw; V->sc = S.sc; V->sh = S.sh; V->mean_w = S.mean_w; V->rstd_w = S.rstd_w;
    V->attn_drop_p = B->attn_drop_p; V->attn_drop_seed = B->attn_drop_seed;
}

}  // namespace

#define RUN(call)                        \
    do {                                 \
        int rc_ = (call);                \
        if (rc_ != PTV2_OK) return rc_;  \
    } while (0)

extern "C" size_t ptv2_block_saved_bytes(int n, int k, int c, int g) {
    if (n < 0 || k < 1 || c < 1 || g < 1) return 0;
    return carve_saved(nullptr, n, k, c, g).bytes + 256;
}

extern "C" size_t ptv2_block_workspace_bytes(int n, int k, int c, int g) {
    if (n < 0 || k < 1 || c < 1 || g < 1) return 0;
    return carve_work(nullptr, n, k, c, g).bytes + 256;
}

extern "C" int ptv2_block_param_layout(int c, int g, long long *offsets) {
    if (c < 1 || g < 1 || !offsets) return PTV2_ERR_ARG;
    long long sz[PTV2_BLK_NPARAM];
    param_sizes(c, g, sz);
    long long off = 0;
    for (int i = 0; i < PTV2_BLK_NPARAM; ++i) {
        offsets[i] = off;
        off += (sz[i] + 3) & ~3LL;  // float4-aligned slots
    }
    offsets[PTV2_BLK_NPARAM] = off;
    return PTV2_OK;
}

static bool use_batch(const ptv2_block *B, int i) { return B->training || !B->run_mean[i] || !B->run_var[i]; }

// statistics of BatchNorm `i` (input h, (n,c)) -> S.mean / S.rstd / S.sc / S.sh: from the producing GEMM's epilogue
// records (`part` != NULL), from a pass over h (`part` == NULL, batch statistics), or from the running buffers (eval)
int bn_tiles_finalize_rb(int n, int c, int rb, float *part, const float *gamma, const float *beta, float *mean, float *rstd, float *sc,
                         float *sh, float *running_mean, float *running_var, long long *num_batches_tracked, float eps, float momentum,
                         void *stream);
int bn_tiles_finalize_pair(int n, int c, float *const *part, const float *const *gamma, const float *const *beta,
                           float *const *mean, float *const *rstd, float *const *sc, float *const *sh, float *const *running_mean,
                           float *const *running_var, long long *const *num_batches_tracked, float eps, float momentum,
                           void *stream, int rb);
// gemm.hip: rows per statistics / reduce record the fused row GEMMs write for a shape (16: the deep levels' k-split kernel),
// and the switch that tells them this caller reads either
int rows_gemm_record_rows(int m, int n, int k);
void ptv2_gemm_allow_rb16(int on);
namespace {
struct GemmRb16Scope {
    GemmRb16Scope() { ptv2_gemm_allow_rb16(1); }
    ~GemmRb16Scope() { ptv2_gemm_allow_rb16(0); }
};
}  // namespace

static int bn_prepare(const ptv2_block *B, int i, const float *h, float *part, const float *gamma, const float *beta,
                      const Saved &S, const Work &W, void *stream, int rb = 64) {
    if (use_batch(B, i)) {
        const bool track = B->training && B->run_mean[i] && B->run_var[i];
        float *rm = track ? B->run_mean[i] : nullptr, *rv = track ? B->run_var[i] : nullptr;
        long long *nb = track ? B->batches[i] : nullptr;
        if (part && rb != 64)
            return bn_tiles_finalize_rb(B->n, B->c, rb, part, gamma, beta, S.mean[i], S.rstd[i], S.bsc[i], S.bsh[i], rm, rv, nb, B->eps,
                                        B->momentum, stream);
        if (part)
            return bn_tiles_finalize_hip_launcher(B->n, B->c, part, gamma, beta, S.mean[i], S.rstd[i], S.bsc[i], S.bsh[i], rm, rv,
                                                  nb, B->eps, B->momentum, stream);
        return bn_stats_affine_hip_launcher(B->n, B->c, h, gamma, beta, S.mean[i], S.rstd[i], S.bsc[i], S.bsh[i], rm, rv, nb, B->eps,
                                            B->momentum, W.dense, W.dense_bytes, stream);
    }
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(divup(B->c, 256)), dim3(256), 0, (hipStream_t)stream, B->c,
                       (const float *)B->run_mean[i], (const float *)B->run_var[i], B->eps, gamma, beta, S.mean[i], S.rstd[i],
                       S.bsc[i], S.bsh[i]);
    return PTV2_OK;
}

int gva_fold_forward_batched(int count, const ptv2_gva_block *blocks, void *stream);
int bn_tiles_apply_residual(int n, int c, float *part, const float *gamma, const float *beta, float *mean, float *rstd, float *sc,
                            float *sh, float *running_mean, float *running_var, long long *num_batches_tracked, float eps,
                            float momentum, const float *x, const float *residual, const float *rowscale, float *y, void *stream,
                            int rb);
int gva_block_forward_stats(const ptv2_gva_block *B, float *out_stats, int *stats_done, void *workspace, size_t workspace_bytes,
                            void *stream);

// internal to the library (model.hip): the parameter-only folds of `count` Blocks in one launch per 8 Blocks, ahead of the
// forward; the Blocks are then run inside ptv2_gva_set_prefolded(1)
int ptv2_blocks_fold_forward(int count, const ptv2_block *blocks, void *stream) {
    if (count < 0 || count > PTV2_MAX_BLOCKS || (count && !blocks)) return PTV2_ERR_ARG;
    ptv2_gva_block V[PTV2_MAX_BLOCKS];
    for (int i = 0; i < count; ++i) {
        const ptv2_block *B = blocks + i;
        if (!args_ok(B)) return PTV2_ERR_ARG;
        const Saved S = carve_saved(B->saved, B->n, B->k, B->c, B->g);
        if (B->saved_bytes < S.bytes) return PTV2_ERR_WORKSPACE;
        fill_gva(B, S, &V[i]);
    }
    return gva_fold_forward_batched(count, V, stream);
}

extern "C" int ptv2_block_forward_hip_launcher(const ptv2_block *B, void *workspace, size_t workspace_bytes, void *stream) {
    if (!args_ok(B)) return PTV2_ERR_ARG;
    const PtvMatmulScope precision(B->matmul_bf16);
    const int n = B->n, k = B->k, c = B->c, g = B->g;
    const Saved S = carve_saved(B->saved, n, k, c, g);
    if (B->saved_bytes < S.bytes) return PTV2_ERR_WORKSPACE;
    const Work W = carve_work(workspace, n, k, c, g);
    if (!workspace || workspace_bytes < W.bytes) return PTV2_ERR_WORKSPACE;
    const float *const *P = B->param;
    const GemmRb16Scope rb16;                          // this function reads the row GEMMs' records at either granularity
    const int rb = rows_gemm_record_rows(n, c, c);     // (every Linear of a Block is (n, c) x (c, c))
    float *st_h1 = use_batch(B, 0) ? W.stat[0] : nullptr, *st_hq = use_batch(B, 1) ? W.stat[1] : nullptr;
    float *st_hk = use_batch(B, 2) ? W.stat[2] : nullptr, *st_h3 = use_batch(B, 6) ? W.stat[3] : nullptr;
    // fc1 (+ statistics of h1) -> norm1
    {
        const float *xs[1] = {B->x}, *ws[1] = {P[PTV2_BLK_FC1_W]};
        float *ys[1] = {S.h1}, *sts[1] = {st_h1};
        RUN(rows_gemm_fused_hip_launcher(n, c, c, 1, 0, xs, ws, 0, nullptr, ys, 0, nullptr, nullptr, sts, stream));
    }
    RUN(bn_prepare(B, 0, S.h1, st_h1, P[PTV2_BLK_N1_G], P[PTV2_BLK_N1_B], S, W, stream, rb));
    // linear_q / linear_k / linear_v on f1 = ReLU(BN1(h1)) (applied on the operand load), statistics of hq, hk
    {
        const float *xs[3] = {S.h1, S.h1, S.h1}, *ws[3] = {P[PTV2_BLK_Q_W], P[PTV2_BLK_K_W], P[PTV2_BLK_V_W]};
        const float *bs[3] = {P[PTV2_BLK_Q_B], P[PTV2_BLK_K_B], P[PTV2_BLK_V_B]};
        float *ys[3] = {S.hq, S.hk, S.v}, *sts[3] = {st_hq, st_hk, nullptr};
        RUN(rows_gemm_fused_hip_launcher(n, c, c, 3, 0, xs, ws, 0, bs, ys, 0, S.bsc[0], S.bsh[0], sts, stream));
    }
    if (st_hq && st_hk) {  // both from their GEMM records: one launch finishes the two BatchNorms
        const bool tq = B->training && B->run_mean[1] && B->run_var[1], tk = B->training && B->run_mean[2] && B->run_var[2];
        float *parts[2] = {st_hq, st_hk}, *means[2] = {S.mean[1], S.mean[2]}, *rstds[2] = {S.rstd[1], S.rstd[2]};
        float *scs[2] = {S.bsc[1], S.bsc[2]}, *shs[2] = {S.bsh[1], S.bsh[2]};
        const float *gs[2] = {P[PTV2_BLK_QN_G], P[PTV2_BLK_KN_G]}, *bs[2] = {P[PTV2_BLK_QN_B], P[PTV2_BLK_KN_B]};
        float *rms[2] = {tq ? B->run_mean[1] : nullptr, tk ? B->run_mean[2] : nullptr};
        float *rvs[2] = {tq ? B->run_var[1] : nullptr, tk ? B->run_var[2] : nullptr};
        long long *nbs[2] = {tq ? B->batches[1] : nullptr, tk ? B->batches[2] : nullptr};
        RUN(bn_tiles_finalize_pair(n, c, parts, gs, bs, means, rstds, scs, shs, rms, rvs, nbs, B->eps, B->momentum, stream, rb));
    } else {
        RUN(bn_prepare(B, 1, S.hq, st_hq, P[PTV2_BLK_QN_G], P[PTV2_BLK_QN_B], S, W, stream, rb));
        RUN(bn_prepare(B, 2, S.hk, st_hk, P[PTV2_BLK_KN_G], P[PTV2_BLK_KN_B], S, W, stream, rb));
    }
    // grouped vector attention (q, k enter as hq, hk + folded affine)
    ptv2_gva_block V;
    fill_gva(B, S, &V);
    // (the attention's last stage leaves the tile statistics of its output where the matrix-core form of it runs: h1's
    // record buffer is free by now; attn_stats = rows per record, 0: none)
    int attn_stats = 0;
    RUN(gva_block_forward_stats(&V, use_batch(B, 5) ? W.stat[0] : nullptr, &attn_stats, W.gva, W.gva_bytes, stream));
    // norm2 (statistics from those records, else by a pass over attn) -> fc3 on f2 = ReLU(BN2(attn)) (+ statistics of h3)
    // -> norm3 -> tail
    RUN(bn_prepare(B, 5, S.attn, attn_stats ? W.stat[0] : nullptr, P[PTV2_BLK_N2_G], P[PTV2_BLK_N2_B], S, W, stream,
                   attn_stats ? attn_stats : 64));
    {
        const float *xs[1] = {S.attn}, *ws[1] = {P[PTV2_BLK_FC3_W]};
        float *ys[1] = {S.h3}, *sts[1] = {st_h3};
        RUN(rows_gemm_fused_hip_launcher(n, c, c, 1, 0, xs, ws, 0, nullptr, ys, 0, S.bsc[5], S.bsh[5], sts, stream));
    }
    // norm3 + tail: at the deep levels the apply kernel merges the tile records itself (one launch instead of two)
    int tail_done = 0;
    if (st_h3) {
        const bool track = B->training && B->run_mean[6] && B->run_var[6];
        tail_done = bn_tiles_apply_residual(n, c, st_h3, P[PTV2_BLK_N3_G], P[PTV2_BLK_N3_B], S.mean[6], S.rstd[6], S.bsc[6], S.bsh[6],
                                            track ? B->run_mean[6] : nullptr, track ? B->run_var[6] : nullptr,
                                            track ? B->batches[6] : nullptr, B->eps, B->momentum, S.h3, B->x, B->rowscale, B->y, stream, rb);
    }
    if (!tail_done) {
        RUN(bn_prepare(B, 6, S.h3, st_h3, P[PTV2_BLK_N3_G], P[PTV2_BLK_N3_B], S, W, stream, rb));
        RUN(bn_apply_residual_hip_launcher(n, c, S.h3, S.mean[6], S.rstd[6], P[PTV2_BLK_N3_G], P[PTV2_BLK_N3_B], B->x, B->rowscale,
                                           B->y, stream));
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int ptv2_block_backward_hip_launcher(const ptv2_block *B, const ptv2_block_grads *G, void *workspace,
                                                size_t workspace_bytes, void *stream) {
    if (!args_ok(B) || !G || !G->gy || !G->gx) return PTV2_ERR_ARG;
    const PtvMatmulScope precision(B->matmul_bf16);
    const int n = B->n, k = B->k, c = B->c, g = B->g;
    const Saved S = carve_saved(B->saved, n, k, c, g);
    if (B->saved_bytes < S.bytes) return PTV2_ERR_WORKSPACE;
    const Work W = carve_work(workspace, n, k, c, g);
    if (!workspace || workspace_bytes < W.bytes) return PTV2_ERR_WORKSPACE;
    const float *const *P = B->param;
    long long off[PTV2_BLK_NPARAM + 1];
    (void)ptv2_block_param_layout(c, g, off);
    // gradient destinations: slots of one flat buffer (ptv2_block_param_layout), or one pointer per parameter
    auto GP = [&](int i) { return G->gparam ? G->gparam + off[i] : G->gp[i]; };
    auto GPB = [&](int i) { return P[i] ? GP(i) : (float *)nullptr; };  // optional biases
    for (int i = 0; i < PTV2_BLK_NPARAM; ++i)
        if (P[i] && !GP(i)) return PTV2_ERR_ARG;
    int batch[PTV2_BLK_NBN];
    for (int i = 0; i < PTV2_BLK_NBN; ++i) batch[i] = (B->training || !B->run_mean[i] || !B->run_var[i]) ? 1 : 0;
    // gradient tensors; g_h3, g_hq, g_hk, gv, g_h1 are also the gY operands of the five (c,c) weight gradients, which run
    // as ONE launch + one finalize at the end of the block instead of three launches spread along the chain
    float *g_h3 = W.t[0], *g_hq = W.t[1], *g_hk = W.t[2], *gv = W.t[3], *g_h1 = W.t[4], *ta = W.t[5], *tb = W.t[6];
    // inside a model backward those five live in the deferral arena instead (the workspace is the next Block's too) and the
    // launch at the end of this function is filed, to run with every other Block's at the end of the backward (dense.hip)
    // -- and so does g_attn, the gradient of the attention's output (norm2's backward writes it, the grouped projection's weight
    // gradient reads it; tb, its place otherwise, is reused for g_f1 further down)
    float *kept = ptv2_wgrad_defer_active() ? ptv2_wgrad_defer_alloc(6 * (size_t)n * c) : nullptr;
    float *g_attn = tb;
    if (kept) {
        const size_t nc = (size_t)n * c;
        g_h3 = kept; g_hq = kept + nc; g_hk = kept + 2 * nc; gv = kept + 3 * nc; g_h1 = kept + 4 * nc; g_attn = kept + 5 * nc;
    }

    // tail: y = ReLU(x + rowscale * BN3(h3)) -> g_h3, residual gradient straight into gx
    RUN(bn_backward_residual_hip_launcher(n, c, S.h3, G->gy, B->y, B->rowscale, S.mean[6], S.rstd[6], P[PTV2_BLK_N3_G],
                                          batch[6], g_h3, G->gx, GP(PTV2_BLK_N3_G), GP(PTV2_BLK_N3_B), W.dense, W.dense_bytes,
                                          stream));
    // fc3 input gradient -> g_f2 (ta); norm2 + ReLU -> g_attn (tb)
    // (at <= 512 row blocks the GEMM's epilogue leaves the reduce records of the BatchNorm backward that consumes its output)
    const GemmRb16Scope rb16;
    const int rb = rows_gemm_record_rows(n, c, c);
    const int nrb = (n + rb - 1) / rb;   // reduce records the GEMM epilogues leave (per 64 rows, or 16 from the k-split kernel)
    // (the full-resolution level -- 1875 records -- measured the same step with the epilogue records as with the separate reduce:
    // profiles/r06_rejected/bn_small_levers.md)
    const bool epi = nrb <= (rb == 16 ? 2048 : 512);
    if (epi) {
        const float *xs[1] = {g_h3}, *ws[1] = {P[PTV2_BLK_FC3_W]};
        RUN(rows_gemm_bnbwd_hip_launcher(n, c, c, 1, xs, ws, 1, ta, S.attn, S.mean[5], S.rstd[5], P[PTV2_BLK_N2_G],
                                         P[PTV2_BLK_N2_B], 1, W.stat[0], stream));
        RUN(bn_backward_records_hip_launcher(n, c, S.attn, ta, S.mean[5], S.rstd[5], P[PTV2_BLK_N2_G], P[PTV2_BLK_N2_B], 1, batch[5],
                                             g_attn, GP(PTV2_BLK_N2_G), GP(PTV2_BLK_N2_B), W.stat[0], nrb, stream));
    } else {
        RUN(rows_gemm_hip_launcher(n, c, c, g_h3, P[PTV2_BLK_FC3_W], 1, nullptr, ta, 0, stream));
        RUN(bn_backward_hip_launcher(n, c, S.attn, ta, S.mean[5], S.rstd[5], P[PTV2_BLK_N2_G], P[PTV2_BLK_N2_B], 1, batch[5], g_attn,
                                     GP(PTV2_BLK_N2_G), GP(PTV2_BLK_N2_B), W.dense, W.dense_bytes, stream));
    }
    // attention: gq (ta), gk (g_h1's buffer, free until the end of the chain), gv
    float *gq = ta, *gk = g_h1;
    ptv2_gva_block V;
    fill_gva(B, S, &V);
    ptv2_gva_block_grads VG;
    VG.g_out = g_attn; VG.inv_ptr = G->inv_ptr; VG.inv_rows = G->inv_rows;
    VG.gq = gq; VG.gk = gk; VG.gv = gv;
    VG.gWp1 = GP(PTV2_BLK_P1_W); VG.gbp1 = GP(PTV2_BLK_P1_B); VG.ggamma_p = GP(PTV2_BLK_PN_G); VG.gbeta_p = GP(PTV2_BLK_PN_B);
    VG.gWp2 = GP(PTV2_BLK_P2_W); VG.gbp2 = GP(PTV2_BLK_P2_B); VG.gWw1 = GP(PTV2_BLK_W1_W); VG.gbw1 = GP(PTV2_BLK_W1_B);
    VG.ggamma_w = GP(PTV2_BLK_WN_G); VG.gbeta_w = GP(PTV2_BLK_WN_B); VG.gWw2 = GP(PTV2_BLK_W2_W); VG.gbw2 = GP(PTV2_BLK_W2_B);
    if (!G->inv_ptr) (void)ptv2_zero_async(gv, sizeof(float) * (size_t)n * c, (hipStream_t)stream);
    // the attention backward's last launch (the skinny input gradients gk, gq) also leaves the reduce records of the two
    // BatchNorm backwards that consume them (dense.hip skinny_bn_bwd_reduce_kernel): one launch fewer per Block
    if (batch[1] == batch[2]) {
        const float *xs[2] = {S.hk, S.hq}, *gys[2] = {gk, gq}, *ms[2] = {S.mean[2], S.mean[1]}, *rs[2] = {S.rstd[2], S.rstd[1]};
        const float *gs[2] = {P[PTV2_BLK_KN_G], P[PTV2_BLK_QN_G]}, *bs[2] = {P[PTV2_BLK_KN_B], P[PTV2_BLK_QN_B]};
        ptv2_skinny_bn_arm(n, c, xs, gys, ms, rs, gs, bs, 1, W.dense, W.dense_bytes);
    }
    ptv2_wgrad_defer_arm_rs(kept != nullptr);
    const int grc = gva_block_backward_hip_launcher(&V, &VG, W.gva, W.gva_bytes, stream);
    ptv2_wgrad_defer_arm_rs(false);
    ptv2_skinny_bn_disarm();
    RUN(grc);
    // linear_k / linear_q BatchNorm + ReLU
    if (batch[1] == batch[2]) {  // one reduce / finalize / apply for both
        const float *xs[2] = {S.hk, S.hq}, *gys[2] = {gk, gq}, *ms[2] = {S.mean[2], S.mean[1]}, *rs[2] = {S.rstd[2], S.rstd[1]};
        const float *gs[2] = {P[PTV2_BLK_KN_G], P[PTV2_BLK_QN_G]}, *bs[2] = {P[PTV2_BLK_KN_B], P[PTV2_BLK_QN_B]};
        float *gxs[2] = {g_hk, g_hq}, *dgs[2] = {GP(PTV2_BLK_KN_G), GP(PTV2_BLK_QN_G)}, *dbs[2] = {GP(PTV2_BLK_KN_B), GP(PTV2_BLK_QN_B)};
        RUN(bn_backward_pair_hip_launcher(n, c, xs, gys, ms, rs, gs, bs, 1, batch[1], gxs, dgs, dbs, W.dense, W.dense_bytes, stream));
    } else {
        RUN(bn_backward_hip_launcher(n, c, S.hk, gk, S.mean[2], S.rstd[2], P[PTV2_BLK_KN_G], P[PTV2_BLK_KN_B], 1, batch[2], g_hk,
                                     GP(PTV2_BLK_KN_G), GP(PTV2_BLK_KN_B), W.dense, W.dense_bytes, stream));
        RUN(bn_backward_hip_launcher(n, c, S.hq, gq, S.mean[1], S.rstd[1], P[PTV2_BLK_QN_G], P[PTV2_BLK_QN_B], 1, batch[1], g_hq,
                                     GP(PTV2_BLK_QN_G), GP(PTV2_BLK_QN_B), W.dense, W.dense_bytes, stream));
    }
    // g_f1 (tb) = g_hq Wq + g_hk Wk + gv Wv
    // norm1 + ReLU -> g_h1; fc1: gx += g_h1 fc1
    {
        const float *xs[3] = {g_hq, g_hk, gv}, *ws[3] = {P[PTV2_BLK_Q_W], P[PTV2_BLK_K_W], P[PTV2_BLK_V_W]};
        if (epi) {
            RUN(rows_gemm_bnbwd_hip_launcher(n, c, c, 3, xs, ws, 1, tb, S.h1, S.mean[0], S.rstd[0], P[PTV2_BLK_N1_G],
                                             P[PTV2_BLK_N1_B], 1, W.stat[0], stream));
            RUN(bn_backward_records_hip_launcher(n, c, S.h1, tb, S.mean[0], S.rstd[0], P[PTV2_BLK_N1_G], P[PTV2_BLK_N1_B], 1,
                                                 batch[0], g_h1, GP(PTV2_BLK_N1_G), GP(PTV2_BLK_N1_B), W.stat[0], nrb, stream));
        } else {
            float *ys[3] = {tb, nullptr, nullptr};
            RUN(rows_gemm_multi_hip_launcher(n, c, c, 3, 1, xs, ws, 1, nullptr, ys, 0, stream));
            RUN(bn_backward_hip_launcher(n, c, S.h1, tb, S.mean[0], S.rstd[0], P[PTV2_BLK_N1_G], P[PTV2_BLK_N1_B], 1, batch[0], g_h1,
                                         GP(PTV2_BLK_N1_G), GP(PTV2_BLK_N1_B), W.dense, W.dense_bytes, stream));
        }
    }
    RUN(rows_gemm_hip_launcher(n, c, c, g_h1, P[PTV2_BLK_FC1_W], 1, nullptr, G->gx, 1, stream));
    // the five (c,c) weight gradients: fc3 (X = f2 = ReLU(BN2(attn))), linear_q / k / v (X = f1 = ReLU(BN1(h1))), fc1 (X = x)
    {
        const float *gys[5] = {g_h3, g_hq, g_hk, gv, g_h1}, *xs[5] = {S.attn, S.h1, S.h1, S.h1, B->x};
        const float *xsc[5] = {S.bsc[5], S.bsc[0], S.bsc[0], S.bsc[0], nullptr};
        const float *xsh[5] = {S.bsh[5], S.bsh[0], S.bsh[0], S.bsh[0], nullptr};
        float *dws[5] = {GP(PTV2_BLK_FC3_W), GP(PTV2_BLK_Q_W), GP(PTV2_BLK_K_W), GP(PTV2_BLK_V_W), GP(PTV2_BLK_FC1_W)};
        float *dbs[5] = {nullptr, GPB(PTV2_BLK_Q_B), GPB(PTV2_BLK_K_B), GPB(PTV2_BLK_V_B), nullptr};
        ptv2_wgrad_defer_arm(kept != nullptr);
        const int wrc = linear_wgrad_multi_hip_launcher(n, c, c, 5, gys, xs, dws, dbs, xsc, xsh, W.dense, W.dense_bytes, stream);
        ptv2_wgrad_defer_arm(false);
        RUN(wrc);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
