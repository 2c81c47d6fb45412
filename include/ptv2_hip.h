/*
 * include/ptv2_hip.h -- C ABI of libptv2_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the reference's `pointops._C` extension
 * (jihun1998/AO, libs/pointops/src).  Every entry point replaces one
 * `extern "C" ..._cuda_launcher` of the reference (cited per function), keeps
 * its argument order, and appends what a correct stream-ordered library needs:
 *
 *   - sizes the reference left implicit (n, b) where our kernels need them,
 *   - a caller-owned scratch workspace (+ a *_workspace_bytes query) instead of
 *     hidden allocations,
 *   - `void *stream` (a hipStream_t; NULL = the legacy default stream the
 *     reference launches on),
 *   - an int status return (the reference returns void and never checks):
 *       PTV2_OK, PTV2_ERR_ARG, PTV2_ERR_WORKSPACE, PTV2_ERR_LAUNCH.
 *
 * All pointers are device pointers unless stated.  fp32 data, int32 indices.
 * Ownership: the caller owns every buffer; the library keeps no state between
 * calls and is re-entrant across host threads.  Buffers marked [zeroed] must be
 * zero-filled by the caller before the call (the reference has the same
 * contract: libs/pointops/functions/interpolation.py:39, aggregation.py:20,
 * grouping.py:31).
 */
#ifndef PTV2_HIP_H
#define PTV2_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTV2_OK 0
#define PTV2_ERR_ARG 1
#define PTV2_ERR_WORKSPACE 2
#define PTV2_ERR_LAUNCH 3

/* Library / build identification (host only). */
int ptv2_abi_version(void);
const char *ptv2_build_info(void);

/* Optional per-kernel timer (measurement aid, off by default; bench.py's roofline object uses it).
 * enable(1) clears the table and makes every launcher bracket its main kernel with HIP events on the launch
 * stream; read() synchronises on them and returns, for kernel id kid in [0, kernel_count), its name (>= 64
 * bytes), the summed duration in microseconds, the number of launches and the mean algorithmic bytes.
 * select(kid >= 0) restricts the bracketing to one kernel id (so a whole-step timing is not perturbed), select(-1)
 * restores all. */
int ptv2_profile_enable(int on);
int ptv2_profile_select(int kid);
int ptv2_profile_stride(int n); /* bracket every n-th launch only (uniform sample), n >= 1 */
int ptv2_profile_is_on(void);
int ptv2_profile_kernel_count(void);
int ptv2_profile_read(int kid, char *name, double *total_us, long long *launches, double *bytes_per_launch);
/* Inside a graph-issued launcher (below) a bracket is a pair of device time-stamp kernels instead of a pair of HIP events
 * (event-record nodes of a graph cannot be read back); the median duration of an EMPTY such bracket, in microseconds,
 * measured on `stream` (< 0 on failure). */
double ptv2_profile_empty_stamp_us(void *stream, int reps);

/* Graph issue of the model launchers (ao_amd/csrc/graph.hip; host only).  ptv2_model_{forward,backward}_hip_launcher
 * capture their kernel sequence on the caller's stream and enqueue it as ONE hipGraph launch (the previous call's
 * executable graph updated in place with this call's arguments): ~1.7 us of host time per kernel instead of ~4, and no
 * per-kernel doorbell.  Same kernels, same arguments, same order as eager issue.  mode(0) / AO_AMD_GRAPH=0 issues
 * eagerly; mode(-1) only reads the setting.  stats: out[9] = scopes, updated, instantiated, declined (ran eagerly),
 * nodes launched, host us capturing / updating+instantiating / launching / waiting for the GPU to release a graph. */
int ptv2_graph_mode(int on);
int ptv2_graph_stats(double *out, int reset);
int ptv2_graph_reset(void);
/* The weight gradients of the Blocks inside ptv2_model_backward_hip_launcher are filed where they are called and run by ONE
 * launch per kernel form at the end of the backward (ao_amd/csrc/dense.hip: WgradJob; same kernels over longer row chunks: equal
 * to ~2e-6 of a gradient's norm).
 * on = 0: every launch where it is called; 1: deferred (default; AO_AMD_WGRAD_DEFER=0 sets 0); -1: query.  Returns the previous
 * setting. */
int ptv2_wgrad_defer_mode(int on);

/* ------------------------------------------------------------------ kNN --
 * Replaces knn_query_cuda_launcher
 *   (libs/pointops/src/knn_query/knn_query_cuda_kernel.h:13, kernel .cu:60-112)
 * and, with pad_with_start=1, pointops2's knnquery_cuda_launcher
 *   (libs/pointops2/src/knnquery/knnquery_cuda_kernel.cu:65-116).
 * For each of the m rows of new_xyz: the nsample nearest rows of xyz inside the
 * same batch segment, ascending squared distance; idx (m,nsample) int32,
 * dist2 (m,nsample) fp32 SQUARED distances (the python wrapper takes the sqrt,
 * libs/pointops/functions/query.py:24).  Missing neighbours: idx -1 (or the
 * segment start when pad_with_start), dist2 1e10.  Results are bit-identical to
 * the reference's heap algorithm, including its behaviour under tied distances.
 * 1 <= nsample <= 128.  xyz (n,3), new_xyz (m,3), offset/new_offset (b) cumulative.
 * new_xyz == xyz && new_offset == offset selects the self-query fast path.
 */
size_t knn_query_hip_workspace_bytes(int m, int n, int b);
int knn_query_hip_launcher(int m, int nsample, const float *xyz, const float *new_xyz,
                           const int *offset, const int *new_offset, int *idx, float *dist2,
                           int n, int b, int pad_with_start, void *workspace,
                           size_t workspace_bytes, void *stream);
/* The same query with the cell grid shared between calls: grid_mode 0 builds the grid of (xyz, offset) in `workspace` (slot 0),
 * grid_mode 1 reuses the grid an earlier call on the same stream built there for the same xyz / offset / n / b (slot 1 .. 3,
 * one per call that shares the grid); the workspace must be the caller's own, sized for the largest m of the calls. */
int knn_query_grid_hip_launcher(int m, int nsample, const float *xyz, const float *new_xyz,
                                const int *offset, const int *new_offset, int *idx, float *dist2,
                                int n, int b, int pad_with_start, int grid_mode, int slot, void *workspace,
                                size_t workspace_bytes, void *stream);
/* Measurement hook (no reference counterpart; bench.py `ops`): while device_counter != NULL the calling thread's
 * knn_query_hip_launcher calls run the counting twin of the grid query kernel, which adds the number of candidate
 * distances it evaluates to *device_counter (64-bit, device memory, zeroed by the caller).  NULL switches it off. */
int knn_query_count_pairs(unsigned long long *device_counter);

/* ------------------------------------------------------------------ FPS --
 * Replaces farthest_point_sampling_cuda_launcher
 *   (libs/pointops/src/sampling/sampling_cuda_kernel.h:13, kernel .cu:14-171);
 * pointops2: furthestsampling_cuda_launcher (same kernel modulo names).
 * b clouds; n_max = largest cloud size (selects the reference's block size B,
 * cuda_utils.h:11-14, which defines its tie rule); tmp (n) fp32 must be
 * pre-filled with 1e10 (sampling.py:19); idx (new_offset[b-1]) int32 out.
 * n_total = total number of points (offset[b-1]), m_total = new_offset[b-1].
 */
size_t farthest_point_sampling_hip_workspace_bytes(int b, int n_total);
int farthest_point_sampling_hip_launcher(int b, int n_max, const float *xyz, const int *offset,
                                         const int *new_offset, float *tmp, int *idx,
                                         int n_total, int m_total, void *workspace,
                                         size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------- grouping --
 * grouping_{forward,backward}_cuda_launcher
 *   (libs/pointops/src/grouping/grouping_cuda_kernel.h:14-15, .cu:5-40).
 * forward: output[m,s,:] = input[idx[m,s],:]   (idx < 0 -> zeros; the reference reads out of bounds)
 * backward: grad_input[idx[m,s],:] += grad_output[m,s,:]     grad_input (n,c) [zeroed]
 */
int grouping_forward_hip_launcher(int m, int nsample, int c, const float *input, const int *idx,
                                  float *output, void *stream);
int grouping_backward_hip_launcher(int m, int nsample, int c, const float *grad_output,
                                   const int *idx, float *grad_input, void *stream);

/* -------------------------------------------------------- interpolation --
 * interpolation_{forward,backward}_cuda_launcher
 *   (libs/pointops/src/interpolation/interpolation_cuda_kernel.h, .cu:5-47).
 * forward: output[n,:] += sum_i input[idx[n,i],:] * weight[n,i]     output (n,c) [zeroed]
 * backward: grad_input[idx[n,i],:] += grad_output[n,:] * weight[n,i]  grad_input (m,c) [zeroed]
 */
/* weights of the inverse-distance interpolation from knn_query's SQUARED distances (m,k), k <= 8:
 * weight[i,s] = (1 / (sqrt(d2) + 1e-8)) / sum_s(...) (libs/pointops/functions/interpolation.py:13-16); idx (m,k) is
 * updated in place: -1 (coarse segment shorter than k) -> idx + n, the reference's negative indexing (:21). */
int interpolation_weights_hip_launcher(int m, int k, int n, const float *dist2, int *idx, float *weight, void *stream);
int interpolation_forward_hip_launcher(int n, int c, int k, const float *input, const int *idx,
                                       const float *weight, float *output, void *stream);
int interpolation_backward_hip_launcher(int n, int c, int k, const float *grad_output,
                                        const int *idx, const float *weight, float *grad_input,
                                        void *stream);
/* the same gradient as a fixed-order gather over the inverse table of idx (inverse_table_hip_launcher on the (n,k)
 * table; requires m <= n): no float atomics, grad_input fully written (no zero fill needed) */
int interpolation_backward_gather_hip_launcher(int m, int c, int k, const float *grad_output, const int *inv_ptr,
                                               const int *inv_rows, const float *weight, float *grad_input,
                                               void *stream);

/* ---------------------------------------------------------- subtraction --
 * subtraction_{forward,backward}_cuda_launcher
 *   (libs/pointops/src/subtraction/subtraction_cuda_kernel.h, .cu:5-44).
 * forward: output[n,s,:] = input1[n,:] - input2[idx[n,s],:]
 * backward: grad_input1[n,:] += g ; grad_input2[idx[n,s],:] -= g      both (n,c) [zeroed]
 */
int subtraction_forward_hip_launcher(int n, int nsample, int c, const float *input1,
                                     const float *input2, const int *idx, float *output,
                                     void *stream);
int subtraction_backward_hip_launcher(int n, int nsample, int c, const int *idx,
                                      const float *grad_output, float *grad_input1,
                                      float *grad_input2, void *stream);

/* ---------------------------------------------------------- aggregation --
 * aggregation_{forward,backward}_cuda_launcher
 *   (libs/pointops/src/aggregation/aggregation_cuda_kernel.h, .cu:5-53).
 * forward: output[n,c] += sum_s (input[idx[n,s],c] + position[n,s,c]) * weight[n,s,c % w_c]   [zeroed]
 * backward: grad_input (n,c) [zeroed], grad_position (n,ns,c), grad_weight (n,ns,w_c) [zeroed]
 */
int aggregation_forward_hip_launcher(int n, int nsample, int c, int w_c, const float *input,
                                     const float *position, const float *weight, const int *idx,
                                     float *output, void *stream);
int aggregation_backward_hip_launcher(int n, int nsample, int c, int w_c, const float *input,
                                      const float *position, const float *weight, const int *idx,
                                      const float *grad_output, float *grad_input,
                                      float *grad_position, float *grad_weight, void *stream);

/* ------------------------------------------------------------ attention --
 * attention_{relation,fusion}_step_{forward,backward}_cuda_launcher
 *   (libs/pointops/src/attention/attention_cuda_kernel.h, .cu:9-147).
 * relation fwd: output[r,g] += sum_c query[tgt[r],g,c] * key[ref[r],g,c] * weight[c]   (m,g) [zeroed]
 * relation bwd: grad_query, grad_key (n,g,c) [zeroed], grad_weight (c) [zeroed]
 * fusion fwd:   output[tgt[r],g,c] += weight[r,g] * value[ref[r],g,c]                (n,g,c) [zeroed]
 * fusion bwd:   grad_weight (m,g) [zeroed], grad_value (n,g,c) [zeroed]
 */
int attention_relation_step_forward_hip_launcher(int m, int g, int c, const float *query,
                                                 const float *key, const float *weight,
                                                 const int *index_target, const int *index_refer,
                                                 float *output, void *stream);
int attention_relation_step_backward_hip_launcher(int m, int g, int c, const float *query,
                                                  float *grad_query, const float *key,
                                                  float *grad_key, const float *weight,
                                                  float *grad_weight, const int *index_target,
                                                  const int *index_refer, const float *grad_output,
                                                  void *stream);
int attention_fusion_step_forward_hip_launcher(int m, int g, int c, const float *weight,
                                               const float *value, const int *index_target,
                                               const int *index_refer, float *output, void *stream);
int attention_fusion_step_backward_hip_launcher(int m, int g, int c, const float *weight,
                                                float *grad_weight, const float *value,
                                                float *grad_value, const int *index_target,
                                                const int *index_refer, const float *grad_output,
                                                void *stream);

/* ------------------------------------------- fused grouped vector attention --
 * Replaces the eager op sequence of GroupedVectorAttention.forward
 *   (pointcept/models/point_transformer_v2/point_transformer_v2m2_base.py:109-128: two
 *   pointops.grouping calls, linear_p_bias, weight_encoding, softmax, mask, einsum) -- the
 *   "grouped-vector-attention kernel" of BASELINE.json -- with two device stages around
 *   which the host keeps the dense products on rocBLAS (ao_amd/ptv2/gva.py has the algebra).
 * n points, k neighbour slots (power of two <= 64 for the aggregate stage), c channels,
 * g groups (6, 12, 24, 48 or 64), i = c / g.  idx (n,k) int32 with -1 placeholders, coord (n,3).
 *   a (c,3), b (c):   folded Linear(3,c)+BatchNorm of linear_p_bias:  P = ReLU(pos a^T + b)
 *   M (c,g), cW (g):  (Ww1 Wp2)^T and Ww1 bp2 + bw1
 *   kW, qW (n,g):     key / query projected by weight_encoding[0]
 * All reductions are per-block partial sums followed by a fixed-order final sum (bitwise
 * reproducible; no float atomics).  workspace: gva_workspace_bytes(n,k,c,g) bytes.
 */
size_t gva_workspace_bytes(int n, int k, int c, int g);
/* s1[3] = sum pos, s2[9] = sum pos pos^T over all n*k slots (float64, device) */
int gva_pos_stats_hip_launcher(int n, int k, const float *coord, const int *idx, double *s1,
                               double *s2, void *workspace, size_t workspace_bytes, void *stream);
/* the same reduced to what the folds need, in two launches: mu (3) = s1 / (n k), cov (9) = s2 / (n k) - mu mu^T, float64 */
int gva_pos_moments_hip_launcher(int n, int k, const float *coord, const int *idx, double *mu, double *cov,
                                 void *workspace, size_t workspace_bytes, void *stream);
/* W1 (n,k,g) = kW[idx]*mask - qW + P M + cW ; T1[g] = sum W1, T2[g] = sum W1^2 (float64) */
int gva_logits_forward_hip_launcher(int n, int k, int c, int g, const float *kW, const float *qW,
                                    const float *a, const float *b, const float *M, const float *cW,
                                    const float *coord, const int *idx, float *W1, double *T1,
                                    double *T2, void *workspace, size_t workspace_bytes, void *stream);
/* inverse neighbour table (optional, may be NULL): inv_ptr (n+1), inv_rows: for point j the slots
 * r = n'*k + s with idx[r] == j are inv_rows[inv_ptr[j] .. inv_ptr[j+1]) in ascending r.  With it the
 * scatter-adds (grad kW, grad v) are fixed-order gathers; without it they fall back to float atomics. */
/* given gW1 (n,k,g), gT1, gT2 (g, float64): gkW (n,g) [zeroed], gqW (n,g), ga (c,3), gb (c), gM (c,g), gcW (g) */
int gva_logits_backward_hip_launcher(int n, int k, int c, int g, const float *a, const float *b,
                                     const float *M, const float *coord, const int *idx,
                                     const float *W1, const float *gW1, const double *gT1,
                                     const double *gT2, const int *inv_ptr, const int *inv_rows,
                                     float *gkW, float *gqW, float *ga, float *gb, float *gM, float *gcW,
                                     void *workspace, size_t workspace_bytes, void *stream);
/* w (n,k,g) = mask * softmax_s(ReLU(sc*W1+sh) Ww2^T + bw2)   [kept by the caller for the backward]
 * out_v (n,c) = sum_s w v[idx];  A (n,g,c) = sum_s w[s,g] P[s,:];  sw (n,g) = sum_s w */
size_t gva_aggregate_workspace_bytes(int n, int k, int c, int g);
int gva_aggregate_forward_hip_launcher(int n, int k, int c, int g, const float *W1, const float *sc,
                                       const float *sh, const float *Ww2, const float *bw2,
                                       const float *v, const float *a, const float *b,
                                       const float *coord, const int *idx, float *out_v, float *A,
                                       float *sw, float *w, void *stream);
/* given w and g_out (n,c), g_A (n,g,c), g_sw (n,g): gW1 (n,k,g), gsc, gsh (g), gWw2 (g,g), gbw2 (g), gv (n,c)
 * [zeroed], ga (c,3), gb (c).  workspace: gva_aggregate_workspace_bytes(n,k,c,g) */
int gva_aggregate_backward_hip_launcher(int n, int k, int c, int g, const float *W1, const float *sc,
                                        const float *sh, const float *Ww2, const float *bw2,
                                        const float *v, const float *a, const float *b,
                                        const float *coord, const int *idx, const float *w,
                                        const float *g_out, const float *g_A, const float *g_sw,
                                        const int *inv_ptr, const int *inv_rows, float *gW1, float *gsc,
                                        float *gsh, float *gWw2, float *gbw2, float *gv, float *ga,
                                        float *gb, void *workspace, size_t workspace_bytes, void *stream);

/* BatchNorm folds of GroupedVectorAttention as single launches (ao_amd/csrc/gva_fold.hip): fold_p maps
 * linear_p_bias[0..1] (Linear(3,c) + BatchNorm over all n*k slots, statistics in closed form from the position
 * moments mu[3], cov[9], float64) to P = ReLU(pos a^T + b); fold_w maps weight_encoding[1] (BatchNorm over the
 * (n*k, g) logits, from their column sums T1, T2) to the affine sc, sh.  training != 0 uses batch statistics and
 * updates running_mean / running_var / num_batches_tracked (pass NULL to skip); rows = n*k. */
int gva_fold_p_forward_hip_launcher(int c, const float *Wp1, const float *bp1, const float *gamma,
                                    const float *beta, const double *mu, const double *cov,
                                    float *running_mean, float *running_var,
                                    long long *num_batches_tracked, int training, double rows, float eps,
                                    float momentum, float *a, float *b, float *rstd, void *stream);
int gva_fold_p_backward_hip_launcher(int c, const float *Wp1, const float *bp1, const float *gamma,
                                     const double *mu, const double *cov, const float *running_mean,
                                     const float *rstd, int training, const float *ga, const float *gb,
                                     float *gWp1, float *gbp1, float *ggamma, float *gbeta, void *stream);
int gva_fold_w_forward_hip_launcher(int g, const double *T1, const double *T2, const float *gamma,
                                    const float *beta, float *running_mean, float *running_var,
                                    long long *num_batches_tracked, int training, double rows, float eps,
                                    float momentum, float *sc, float *sh, double *mean, double *rstd,
                                    void *stream);
int gva_fold_w_backward_hip_launcher(int g, const float *gamma, const double *mean, const double *rstd,
                                     int training, double rows, const float *gsc, const float *gsh,
                                     double *gT1, double *gT2, float *ggamma, float *gbeta, void *stream);

/* grouped positional-bias projection applied after the neighbour sum (linear_p_bias[3], :92,117-119):
 *   out[n,g*I+i] = out_v[n,g*I+i] + sum_c' A[n,g,c'] Wp2[g*I+i,c'] + bp2[g*I+i] sw[n,g],  I = c/g in {2,4,8,16}
 * backward w.r.t. A and sw: g_A (n,g,c), g_sw (n,g) from g_out (n,c) (grad Wp2 / bp2 are dense products the
 * host takes with rocBLAS). */
int gva_peb_forward_hip_launcher(int n, int c, int g, const float *A, const float *Wp2, const float *bp2,
                                 const float *sw, const float *out_v, float *out, void *stream);
int gva_peb_backward_hip_launcher(int n, int c, int g, const float *g_out, const float *Wp2,
                                  const float *bp2, float *g_A, float *g_sw, void *stream);
/* The two stages above and the projection as ONE launch (ao_amd/csrc/gva_fwd_tile.hip; k = 16 and (g, c) one of (12, 96),
 * (24, 192), (48, 384), (64, 512) -- the deep levels of PT-v2m2; PTV2_ERR_ARG otherwise): a workgroup owns 16 points x a block
 * of groups, A = w^T P lives in LDS per 16-channel chunk, Wp2 is streamed once per tile.  Writes w (n,k,g), sw (n,g) and
 * out (n,c) = sum_s w v[idx] + A Wp2^T (grouped) + bp2 sw -- GroupedVectorAttention.forward :122-129 behind the folded
 * positional encoding.  A (n,g,c) is written only when A != NULL (the staged backward reads it). */
int gva_attention_forward_hip_launcher(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh,
                                       const float *Ww2, const float *bw2, const float *v, const float *a,
                                       const float *b, const float *coord, const int *idx, const float *Wp2,
                                       const float *bp2, float *w, float *sw, float *out, float *A, void *stream);
/* Its backward as one launch per tile of points + the inverse-table gather of grad v (ao_amd/csrc/gva_bwd_tile.hip; the same
 * shapes): given w (n,k,g) of the forward and g_out (n,c) -- g_A = g_out Wp2 per group is formed in LDS per 16-channel chunk,
 * never in memory -- writes gW1 (n,k,g), gsc, gsh (g), gWw2 (g,g), gbw2 (g), gv (n,c), ga (c,3), gb (c) (the gradients of the
 * folded positional encoding P = ReLU(a . pos + b)).  inv_ptr / inv_rows: the inverse neighbour table (inverse_table);
 * workspace: gva_aggregate_workspace_bytes(n,k,c,g).  grad Wp2 / bp2 (direct parts) are the caller's (a weight gradient). */
int gva_attention_backward_hip_launcher(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh,
                                        const float *Ww2, const float *bw2, const float *v, const float *a,
                                        const float *b, const float *coord, const int *idx, const float *w,
                                        const float *g_out, const float *Wp2, const float *bp2, const int *inv_ptr,
                                        const int *inv_rows, float *gW1, float *gsc, float *gsh, float *gWw2, float *gbw2,
                                        float *gv, float *ga, float *gb, void *workspace, size_t workspace_bytes,
                                        void *stream);

/* ------------------------------------------ whole attention block, one call --
 * GroupedVectorAttention.forward / backward (point_transformer_v2m2_base.py:103-129) behind ONE launcher each:
 * the stage launchers above are enqueued back to back from native code, so the host pays one call instead of
 * ~35 python-level ops per block.  All tensors are caller-owned; the forward fills the
 * "saved" fields, which the backward reads.  q, k are the outputs of linear_q / linear_k (after BN+ReLU), v of
 * linear_v; mu[3], cov[9] are the float64 position moments of the neighbour table (gva_pos_stats); bn_*
 * running statistics may be NULL (no tracking); training selects batch statistics. */
typedef struct ptv2_gva_block {
    int n, k, c, g, training;
    float eps_p, momentum_p, eps_w, momentum_w;
    /* inputs */
    const float *q, *key, *v, *coord;
    const int *idx;
    const double *mu, *cov;
    /* parameters: linear_p_bias[0], its BatchNorm, linear_p_bias[3], weight_encoding[0], its BatchNorm, [3] */
    const float *Wp1, *bp1, *gamma_p, *beta_p, *Wp2, *bp2, *Ww1, *bw1, *gamma_w, *beta_w, *Ww2, *bw2;
    float *run_mean_p, *run_var_p, *run_mean_w, *run_var_w;
    long long *batches_p, *batches_w;
    /* output */
    float *out;                                   /* (n,c) */
    /* saved by forward for backward */
    float *a, *b, *rstd_p, *M, *cW;               /* (c,3) (c) (c) (c,g) (g) */
    float *kW, *qW, *W1, *w, *A, *sw, *sc, *sh;   /* (n,g) (n,g) (n,k,g) (n,k,g) (n,g,c) (n,g) (g) (g) */
    double *mean_w, *rstd_w;                      /* (g) (g) */
    /* optional (all four or none): q / key are the pre-BatchNorm outputs of linear_q[0] / linear_k[0] and the
     * BatchNorm + ReLU is applied as ReLU(x * sc + sh) on the operand load of the consumers (block.hip) */
    const float *q_sc, *q_sh, *k_sc, *k_sh;       /* (c) each */
    /* attention dropout (attn_drop of the reference, :101,122), training only: every softmax weight (point, slot, group) is
     * multiplied by Bernoulli(1 - p) / (1 - p); the mask is a counter-based hash of (attn_drop_seed, element index) that
     * forward and backward kernels evaluate again (never stored).  The backward must be given the forward's seed.  p = 0: off */
    float attn_drop_p;
    unsigned attn_drop_seed;
} ptv2_gva_block;

typedef struct ptv2_gva_block_grads {
    const float *g_out;                           /* (n,c) */
    const int *inv_ptr, *inv_rows;                /* inverse neighbour table (NULL only for shapes without a fused backward:
                                                   * (6,48) and the deep-level instances (12,96) (24,192) (48,384) need it) */
    float *gq, *gk, *gv;                          /* (n,c) each; gv [zeroed] when inv_ptr == NULL */
    float *gWp1, *gbp1, *ggamma_p, *gbeta_p, *gWp2, *gbp2, *gWw1, *gbw1, *ggamma_w, *gbeta_w, *gWw2, *gbw2;
} ptv2_gva_block_grads;

size_t gva_block_workspace_bytes(int n, int k, int c, int g);
int gva_block_forward_hip_launcher(const ptv2_gva_block *blk, void *workspace, size_t workspace_bytes,
                                   void *stream);
int gva_block_backward_hip_launcher(const ptv2_gva_block *blk, const ptv2_gva_block_grads *grads,
                                    void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------- grid pooling --
 * Device pieces of GridPool.forward (point_transformer_v2m2_base.py:244-269), which the reference
 * builds from torch_scatter.segment_csr calls (third party, not vendored):
 *   segment_minmax: per-cloud coordinate min/max, lo/hi (b,3)                         (:249-253)
 *   pool_max:       out[j,:] = max over rows order[idx_ptr[j]..idx_ptr[j+1]) of feat; arg = winning row
 *                   (first wins ties); backward scatters grad_out to grad_feat[arg] [zeroed]   (:266)
 */
/*   grid_pool:     the whole coordinate half of GridPool.forward (:246-268): voxel ids (grid_cluster formula,
 *                  batch-major), stable sort, cluster ranks, pooled coordinates (mean in ascending point order),
 *                  new offsets.  cluster (n) int64, order (n), idx_ptr (n+1), new_coord (n,3), new_offset (b) are
 *                  sized for the worst case; *n_out (device int) = number of clusters (-1: voxel id overflow).
 *                  sort_path == 0 tabulates the voxel grid (no sort; grids up to 2^23 cells): *n_out == -2 then says the grid
 *                  of this batch is larger and the call has to be repeated with sort_path != 0 (radix sort of the ids).
 *                  Both paths give identical outputs. */
size_t grid_pool_hip_workspace_bytes(int n, int b);
int grid_pool_hip_launcher(int n, int b, const float *coord, const int *offset, float grid_size,
                           long long *cluster, int *order, int *idx_ptr, float *new_coord, int *new_offset,
                           int *n_out, int sort_path, void *workspace, size_t workspace_bytes, void *stream);
/*   inverse_table: CSR inverse of a neighbour table idx (n,k): for point j the slots r with idx[r] == j are
 *                  inv_rows[inv_ptr[j] .. inv_ptr[j+1]) in ascending r; inv_ptr (n+1), inv_rows (n*k). */
size_t inverse_table_hip_workspace_bytes(int n, int k);
int inverse_table_hip_launcher(int n, int k, const int *idx, int *inv_ptr, int *inv_rows, void *workspace,
                               size_t workspace_bytes, void *stream);
/*   inverse_tables: the same for up to PTV2_INVERSE_MAX_JOBS tables at once (all neighbour / interpolation tables of a
 *                  scene) in the same five launches (ao_amd/csrc/inverse.hip: counting sort, no library sort).
 *                  `jobs` is a HOST array, read during the call. */
#define PTV2_INVERSE_MAX_JOBS 16
typedef struct ptv2_inverse_job {
    int n, k;          /* table (n, k), entries in [-1, n) */
    const int *idx;
    int *inv_ptr;      /* (n + 1) */
    int *inv_rows;     /* (n * k) */
} ptv2_inverse_job;
size_t inverse_tables_hip_workspace_bytes(int count, const ptv2_inverse_job *jobs);
int inverse_tables_hip_launcher(int count, const ptv2_inverse_job *jobs, void *workspace, size_t workspace_bytes,
                                void *stream);
size_t segment_minmax_hip_workspace_bytes(int b);
int segment_minmax_hip_launcher(int b, const float *xyz, const int *offset, float *lo, float *hi,
                                void *workspace, size_t workspace_bytes, void *stream);
int pool_max_forward_hip_launcher(int n_out, int c, const float *feat, const int *order,
                                  const int *idx_ptr, float *out, int *arg, void *stream);
int pool_max_backward_hip_launcher(int n_out, int c, const float *grad_out, const int *arg,
                                   float *grad_feat, void *stream);

/*   segment_sum:   grad_coarse[j,:] = sum over rows order[idx_ptr[j]..idx_ptr[j+1]) of grad_fine -- the backward of the
 *                  "map" unpool feat[cluster] (:305-310) in the CSR order of the pooling (no index_put atomics) */
int segment_sum_hip_launcher(int n_out, int c, const float *grad_fine, const int *order, const int *idx_ptr,
                             float *grad_coarse, void *stream);

/* ------------------------------------------------ per-point dense layers --
 * PointBatchNorm on (N,C) rows (point_transformer_v2m2_base.py:26-45: nn.BatchNorm1d) with an optional
 * fused ReLU, and the weight/bias gradient of nn.Linear as a split-K reduction.  c % 4 == 0, c <= 1024.
 *   bn_stats:    batch mean / rstd (biased variance) of x (n,c); when running_mean != NULL also the
 *                momentum update of running_mean / running_var (unbiased) and ++*num_batches_tracked
 *   bn_apply:    y = (x - mean) * rstd * gamma + beta, then max(.,0) if relu
 *   bn_backward: gx, dgamma, dbeta from gy (n,c); relu != 0 masks gy where the forward output was <= 0;
 *                training == 0 treats mean / rstd as constants (eval mode)
 *   linear_wgrad: dW (cout,cin) = gY^T X, db (cout) = column sums of gY (db may be NULL)
 */
size_t dense_workspace_bytes(int n, int cout, int cin);
int bn_stats_hip_launcher(int n, int c, const float *x, float *mean, float *rstd, float *running_mean,
                          float *running_var, long long *num_batches_tracked, float eps, float momentum,
                          void *workspace, size_t workspace_bytes, void *stream);
/* training-mode forward as one call: statistics + running buffers + y = [ReLU](BN(x)), or with residual != NULL the
 * Block tail y = ReLU(residual + rowscale * BN(x)) */
int bn_forward_hip_launcher(int n, int c, const float *x, const float *gamma, const float *beta, int relu,
                            float *mean, float *rstd, float *running_mean, float *running_var,
                            long long *num_batches_tracked, float eps, float momentum, const float *residual,
                            const float *rowscale, float *y, void *workspace, size_t workspace_bytes, void *stream);
int bn_apply_hip_launcher(int n, int c, const float *x, const float *mean, const float *rstd,
                          const float *gamma, const float *beta, int relu, float *y, void *stream);
int bn_backward_hip_launcher(int n, int c, const float *x, const float *gy, const float *mean,
                             const float *rstd, const float *gamma, const float *beta, int relu,
                             int training, float *gx, float *dgamma, float *dbeta, void *workspace,
                             size_t workspace_bytes, void *stream);
/* two independent BatchNorm backwards of one shape (arrays of 2 pointers each) in the three launches of one;
 * workspace: dense_workspace_bytes(n, 2 * c, c) */
int bn_backward_pair_hip_launcher(int n, int c, const float *const *x, const float *const *gy,
                                  const float *const *mean, const float *const *rstd, const float *const *gamma,
                                  const float *const *beta, int relu, int training, float *const *gx,
                                  float *const *dgamma, float *const *dbeta, void *workspace, size_t workspace_bytes,
                                  void *stream);
/* The reduce pass of a BatchNorm (+ ReLU) backward inside the GEMM that produces its incoming gradient:
 *   rows_gemm_bnbwd: Y (m,n) = sum_i X[i] op(W[i]) (count <= 3 pairs, as rows_gemm_multi with sum != 0), and `records`
 *     receives ceil(m / 64) records of [2][n] floats: column sums of g' and of g' * xhat over the record's 64 rows, with
 *     g' = Y masked by ReLU(BN(bn_x)) when relu != 0 and xhat = (bn_x - bn_mean) * bn_rstd;
 *   bn_backward_records: bn_backward without its reduce launch -- merges nrec such records into dgamma / dbeta (fixed
 *     order, float64) and applies.  Together: 3 launches instead of 4, two (m,n) tensor reads less. */
int rows_gemm_bnbwd_hip_launcher(int m, int n, int k, int count, const float *const *X, const float *const *W,
                                 int w_kmajor, float *Y, const float *bn_x, const float *bn_mean, const float *bn_rstd,
                                 const float *bn_gamma, const float *bn_beta, int relu, float *records, void *stream);
int bn_backward_records_hip_launcher(int n, int c, const float *x, const float *gy, const float *mean,
                                     const float *rstd, const float *gamma, const float *beta, int relu,
                                     int training, float *gx, float *dgamma, float *dbeta, const float *records,
                                     int nrec, void *stream);
/* Block tail fused into the Block's last BatchNorm (point_transformer_v2m2_base.py:174-176):
 * y = ReLU(residual + rowscale[n] * BN(x)); rowscale (n) = per-point DropPath factor or NULL.  Backward returns
 * the BN input gradient gx, the residual gradient g_residual = gy * (y > 0), dgamma, dbeta. */
int bn_apply_residual_hip_launcher(int n, int c, const float *x, const float *mean, const float *rstd,
                                   const float *gamma, const float *beta, const float *residual,
                                   const float *rowscale, float *y, void *stream);
int bn_backward_residual_hip_launcher(int n, int c, const float *x, const float *gy, const float *y,
                                      const float *rowscale, const float *mean, const float *rstd,
                                      const float *gamma, int training, float *gx, float *g_residual,
                                      float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                                      void *stream);
int linear_wgrad_hip_launcher(int n, int cout, int cin, const float *gY, const float *X, float *dW,
                              float *db, void *workspace, size_t workspace_bytes, void *stream);
/* count (<= 6) weight gradients of one shape in one launch: dW[i] = gY[i]^T X[i], db[i] = column sums (db or db[i]
 * may be NULL); workspace: dense_workspace_bytes(n, count * cout, cin) */
int linear_wgrad_multi_hip_launcher(int n, int cout, int cin, int count, const float *const *gY,
                                    const float *const *X, float *const *dW, float *const *db,
                                    const float *const *xsc, const float *const *xsh, void *workspace,
                                    size_t workspace_bytes, void *stream);
/* xsc[i] / xsh[i] (cin) != NULL (arrays may be NULL): the X operand of product i is ReLU(x * xsc + xsh), i.e. the
 * BatchNorm + ReLU in front of that Linear applied on the operand load instead of in a pass of its own. */
/* skinny projection y (n,cout) = x (n,cin) W^T (cout,cin), cout <= 64, cin % 4 == 0, and its input gradient
 * gx = gy W (the weight gradient is linear_wgrad) */
int skinny_linear_forward_hip_launcher(int n, int cin, int cout, const float *x, const float *W, float *y,
                                       void *stream);
/* the same with the input passed through ReLU(x * xsc + xsh) first (xsc, xsh (cin), both or neither) */
int skinny_linear_forward_xf_hip_launcher(int n, int cin, int cout, const float *x, const float *W, const float *xsc,
                                          const float *xsh, float *y, void *stream);
int skinny_linear_backward_hip_launcher(int n, int cin, int cout, const float *gy, const float *W,
                                        float *gx, void *stream);
/* batched / strided form: dW[b][o][i] = sum_n gY[n*ldy + b*sy + o] * X[n*ldx + b*sx + i], b < batch
 * (workspace: dense_workspace_bytes(n, batch*cout, cin)) */
int linear_wgrad_strided_hip_launcher(int n, int cout, int cin, int batch, const float *gY,
                                      long long ldy, long long sy, const float *X, long long ldx,
                                      long long sx, float *dW, float *db, void *workspace,
                                      size_t workspace_bytes, void *stream);

/* -------------------------------------------------- fp32 row GEMM (MFMA) --
 * Y (m,n) [+]= X (m,k) op(W) + bias: the nn.Linear products of a Block on V_MFMA_F32_16X16X4_F32 (true fp32).
 * w_kmajor == 0: W is (n,k) row-major, y = x W^T (forward); w_kmajor != 0: W is (k,n) row-major, gx = gy W
 * (input gradient, same weight matrix).  n % 4 == 0, k % 4 == 0, bias (n) or NULL, accumulate != 0 adds onto Y. */
int rows_gemm_hip_launcher(int m, int n, int k, const float *X, const float *W, int w_kmajor,
                           const float *bias, float *Y, int accumulate, void *stream);

/* count (<= 3) products of one shape in one launch.  sum == 0: Y[i] = X[i] op(W[i]) + bias[i] (e.g. the q, k, v
 * projections of one input); sum != 0: Y[0] (+)= sum_i X[i] op(W[i]) (the input gradient through all three). */
int rows_gemm_multi_hip_launcher(int m, int n, int k, int count, int sum, const float *const *X,
                                 const float *const *W, int w_kmajor, const float *const *bias, float *const *Y,
                                 int accumulate, void *stream);

/* BatchNorm fused into the Linear on either side of it (block.hip):
 *   rows_gemm_fused: as rows_gemm_multi; xsc / xsh (k) != NULL: the X operand is ReLU(x * xsc + xsh) (normalise + ReLU of
 *     the BatchNorm in front, folded to one multiply-add per element); stats != NULL and stats[i] != NULL: per-row-block
 *     column statistics of Y[i], ceil(m / 64) records of [2][n] floats (sum; sum of squares about the block mean)
 *   bn_tiles_finalize: merges those records (parallel-variance identity, float64) into mean / rstd, updates the running
 *     buffers, and (sc != NULL) emits the folded affine sc = rstd * gamma, sh = beta - mean * sc
 *   bn_stats_affine: bn_stats that also emits sc, sh */
int rows_gemm_fused_hip_launcher(int m, int n, int k, int count, int sum, const float *const *X,
                                 const float *const *W, int w_kmajor, const float *const *bias, float *const *Y,
                                 int accumulate, const float *xsc, const float *xsh, float *const *stats, void *stream);
size_t bn_tiles_floats(int n, int c); /* floats to reserve for one statistics record buffer */
int bn_tiles_finalize_hip_launcher(int n, int c, float *part, const float *gamma, const float *beta, float *mean,
                                   float *rstd, float *sc, float *sh, float *running_mean, float *running_var,
                                   long long *num_batches_tracked, float eps, float momentum, void *stream);
int bn_stats_affine_hip_launcher(int n, int c, const float *x, const float *gamma, const float *beta, float *mean,
                                 float *rstd, float *sc, float *sh, float *running_mean, float *running_var,
                                 long long *num_batches_tracked, float eps, float momentum, void *workspace,
                                 size_t workspace_bytes, void *stream);

/* ------------------------------------------------ whole Block, one call --
 * Block.forward / backward (point_transformer_v2m2_base.py:131-177: fc1, norm1, GroupedVectorAttention with
 * linear_q/k/v, norm2, fc3, norm3, DropPath, residual, ReLU) behind ONE launcher per direction: ~30 forward and
 * ~75 backward kernels are enqueued on the caller's stream from native code.  The step was host-bound on the
 * python / autograd dispatch of those ops; a Block now costs two host calls.
 * param[] follows PTV2_BLK_* (reference state_dict names in the comments); biases may be NULL (qkv_bias=False).
 * bn index: 0 norm1, 1 linear_q[1], 2 linear_k[1], 3 linear_p_bias[1], 4 weight_encoding[1], 5 norm2, 6 norm3.
 * `saved` (ptv2_block_saved_bytes) is written by the forward and read by the backward together with y.
 * gparam is one flat float buffer laid out by ptv2_block_param_layout (offsets[i] of param i, offsets[30] = total). */
enum {
    PTV2_BLK_FC1_W = 0,                                   /* fc1.weight (c,c) */
    PTV2_BLK_N1_G, PTV2_BLK_N1_B,                         /* norm1.norm.{weight,bias} */
    PTV2_BLK_Q_W, PTV2_BLK_Q_B, PTV2_BLK_QN_G, PTV2_BLK_QN_B,   /* attn.linear_q.0.*, attn.linear_q.1.norm.* */
    PTV2_BLK_K_W, PTV2_BLK_K_B, PTV2_BLK_KN_G, PTV2_BLK_KN_B,   /* attn.linear_k.* */
    PTV2_BLK_V_W, PTV2_BLK_V_B,                           /* attn.linear_v.* */
    PTV2_BLK_P1_W, PTV2_BLK_P1_B, PTV2_BLK_PN_G, PTV2_BLK_PN_B, PTV2_BLK_P2_W, PTV2_BLK_P2_B, /* attn.linear_p_bias.{0,1.norm,3} */
    PTV2_BLK_W1_W, PTV2_BLK_W1_B, PTV2_BLK_WN_G, PTV2_BLK_WN_B, PTV2_BLK_W2_W, PTV2_BLK_W2_B, /* attn.weight_encoding.{0,1.norm,3} */
    PTV2_BLK_N2_G, PTV2_BLK_N2_B,                         /* norm2.norm.* */
    PTV2_BLK_FC3_W,                                       /* fc3.weight */
    PTV2_BLK_N3_G, PTV2_BLK_N3_B,                         /* norm3.norm.* */
    PTV2_BLK_NPARAM
};
#define PTV2_BLK_NBN 7
typedef struct ptv2_block {
    int n, k, c, g, training;
    float eps, momentum;
    const float *x;            /* (n,c) block input, also the residual */
    const float *coord;        /* (n,3) */
    const int *idx;            /* (n,k) neighbour table, -1 placeholders */
    const double *mu, *cov;    /* position moments of the table (gva_pos_stats) */
    const float *rowscale;     /* (n) per-point DropPath factor or NULL */
    const float *param[PTV2_BLK_NPARAM];
    float *run_mean[PTV2_BLK_NBN], *run_var[PTV2_BLK_NBN];
    long long *batches[PTV2_BLK_NBN];
    float *y;                  /* (n,c) output */
    void *saved;
    size_t saved_bytes;
    int matmul_bf16;           /* != 0: the Linear products (fc1, fc3, q/k/v and their input / weight gradients) run on the
                                * bf16 matrix cores: operands rounded to bf16, fp32 accumulation, fp32 in memory -- the
                                * arithmetic torch.autocast(bfloat16) gives nn.Linear in the reference trainer */
    float attn_drop_p;         /* attention dropout of this Block's GroupedVectorAttention (see ptv2_gva_block), 0: off */
    unsigned attn_drop_seed;   /* the same value in the forward and the backward call of a step */
} ptv2_block;
typedef struct ptv2_block_grads {
    const float *gy;               /* (n,c) */
    const int *inv_ptr, *inv_rows; /* inverse neighbour table (may be NULL) */
    float *gx;                     /* (n,c) */
    float *gparam;                 /* flat parameter gradients (layout: ptv2_block_param_layout), or NULL: */
    float *gp[PTV2_BLK_NPARAM];    /* ... one destination per parameter (float4-aligned; NULL for absent biases) */
} ptv2_block_grads;
size_t ptv2_block_saved_bytes(int n, int k, int c, int g);
size_t ptv2_block_workspace_bytes(int n, int k, int c, int g);
int ptv2_block_param_layout(int c, int g, long long *offsets /* [PTV2_BLK_NPARAM + 1] */);
int ptv2_block_forward_hip_launcher(const ptv2_block *blk, void *workspace, size_t workspace_bytes, void *stream);
int ptv2_block_backward_hip_launcher(const ptv2_block *blk, const ptv2_block_grads *grads, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* ------------------------------------------------ whole model, one call --
 * PointTransformerV2.forward / backward (point_transformer_v2m2_base.py:556-576: patch embedding, S encoder stages of
 * GridPool + BlockSequence, S decoder stages of UnpoolWithSkip + BlockSequence, segmentation head) behind ONE launcher
 * per direction.  Every kernel of the step between `feat` and `seg_logits` is enqueued from native code on the caller's
 * stream: no python / autograd dispatch between the stages (the step was host-bound on it: ~9 ms of host time beside
 * ~4 ms of launch calls), no torch glue kernels (zero fills, adds, copies).  Geometry (neighbour tables, pooling CSR,
 * interpolation tables) is the caller's, built once per batch (ao_amd/ptv2/geometry.py); parameter gradients are written
 * straight to the destinations the caller names (e.g. slots of an optimizer's flat gradient buffer).
 *   ptv2_linbn : Sequential(Linear, PointBatchNorm, ReLU) (GridPool.fc/norm :240-242, UnpoolWithSkip.proj /
 *                proj_skip :293-302, GVAPatchEmbed.proj :424-428, seg_head[0:3] :545-550); g* = gradient destinations
 *   level i    : resolution i (0 = input); its link to level i+1 (pooling CSR, cluster map, 3-NN interpolation tables)
 *   seq        : 0 = patch_embed.blocks (level 0); 1+i = enc_stages[i].blocks (level i+1); 1+S+i = dec_stages[i].blocks
 *                (level i); blocks first_block .. first_block+depth-1 of `block[]`
 *   saved      : ptv2_model_saved_bytes() bytes written by the forward, read by the backward. */
#define PTV2_MAX_STAGES 5
#define PTV2_MAX_BLOCKS 40
typedef struct ptv2_linbn {
    int cin, cout;
    const float *w, *b;                /* (cout,cin); (cout) or NULL */
    const float *gamma, *beta;         /* (cout) */
    float *run_mean, *run_var;         /* running buffers (NULL: batch statistics always) */
    long long *batches;
    float *gw, *gb, *ggamma, *gbeta;   /* backward only */
} ptv2_linbn;
typedef struct ptv2_level {
    int n, b;
    const float *coord;                /* (n,3) */
    const int *offset;                 /* (b) */
    const int *order, *idx_ptr;        /* CSR of the pooling to level i+1: (n), (n_{i+1} + 1) */
    const long long *cluster;          /* (n) point -> cluster, "map" unpool */
    const int *up_idx;                 /* (n,3) into level i+1, "interp" unpool */
    const float *up_w;                 /* (n,3) */
    const int *up_inv_ptr, *up_inv_rows; /* inverse table of up_idx: (n_{i+1} + 1), (3 n) */
} ptv2_level;
typedef struct ptv2_seq {
    int level, depth, first_block, c, g, k;
    const int *idx;                    /* (n,k) */
    const double *mu, *cov;            /* position moments of the table */
    const int *inv_ptr, *inv_rows;     /* inverse neighbour table */
} ptv2_seq;
typedef struct ptv2_model_block {
    const float *param[PTV2_BLK_NPARAM];
    float *run_mean[PTV2_BLK_NBN], *run_var[PTV2_BLK_NBN];
    long long *batches[PTV2_BLK_NBN];
    float *gparam[PTV2_BLK_NPARAM];    /* backward only */
    const float *rowscale;             /* (n) DropPath factors of this block or NULL */
    float attn_drop_p;                 /* attention dropout of this block (ptv2_gva_block), 0: off */
    unsigned attn_drop_seed;
} ptv2_model_block;
typedef struct ptv2_model {
    int num_stages, in_channels, num_classes, training, interp; /* interp != 0: "interp" unpool, else "map" */
    float eps, momentum;
    ptv2_level level[PTV2_MAX_STAGES + 1];
    ptv2_seq seq[2 * PTV2_MAX_STAGES + 1];
    int num_blocks;
    ptv2_model_block block[PTV2_MAX_BLOCKS];
    ptv2_linbn embed, down[PTV2_MAX_STAGES], up[PTV2_MAX_STAGES], up_skip[PTV2_MAX_STAGES], head;
    const float *head_w, *head_b;      /* seg_head[3]: (num_classes, c0), (num_classes) */
    float *g_head_w, *g_head_b;        /* backward only */
    const float *feat;                 /* (n0, in_channels) */
    float *logits;                     /* (n0, num_classes) */
    void *saved;
    size_t saved_bytes;
    int matmul_bf16;                   /* as ptv2_block.matmul_bf16, for every Linear of the network */
    int checkpoint;                    /* != 0: activation checkpointing of the Blocks (enable_checkpoint of the reference,
                                        * point_transformer_v2m2_base.py:169-171): the forward keeps only every Block's
                                        * output; all Blocks share ONE saved-activation region, and the backward re-runs a
                                        * Block's forward right before its
                                        * backward.  Logits and gradients bit for bit, ptv2_model_saved_bytes() shrinks; the running
                                        * statistics of the norms inside the attention advance twice per step, as under
                                        * torch.utils.checkpoint in the reference. */
    void *decoder_done_event;          /* backward only, optional hipEvent_t: recorded on `stream` once the head and every
                                        * decoder stage have written their parameter gradients (the tail of the flat
                                        * gradient buffer in module.parameters() order) -- a data-parallel caller starts
                                        * the all-reduce of that half while the encoder's backward still runs */
    void *saved0;                      /* != NULL: the activations of the PREFIX (patch embedding + seq 0: everything that needs
                                        * level 0 only) live here, ptv2_model_prefix_saved_bytes() bytes, and `saved` /
                                        * ptv2_model_saved_bytes() cover the rest alone -- the two halves of a forward issued as
                                        * prefix + rest (below); not with `checkpoint` */
    size_t saved0_bytes;
} ptv2_model;
size_t ptv2_model_saved_bytes(const ptv2_model *m);
size_t ptv2_model_workspace_bytes(const ptv2_model *m);
int ptv2_model_forward_hip_launcher(const ptv2_model *m, void *workspace, size_t workspace_bytes, void *stream);
/* The forward in two calls, for a caller that learns the sizes of levels 1.. only while level 0 is already computing
 * (ao_amd/ptv2/native_model.py: the grid poolings' 4-byte read-backs run on a side stream behind the prefix's launches):
 *   prefix: GVAPatchEmbed (point_transformer_v2m2_base.py:441-444) -- reads level[0], seq[0], embed, block[seq[0] blocks],
 *           feat, saved0 only; the sizes are functions of those fields
 *   rest:   encoder stages, decoder stages, head (:560-576) -- the whole struct, saved0 as the prefix left it
 * Together they enqueue exactly the kernels of ptv2_model_forward_hip_launcher (same results bit for bit). */
size_t ptv2_model_prefix_saved_bytes(const ptv2_model *m);
size_t ptv2_model_prefix_workspace_bytes(const ptv2_model *m);
int ptv2_model_forward_prefix_hip_launcher(const ptv2_model *m, void *workspace, size_t workspace_bytes, void *stream);
int ptv2_model_forward_rest_hip_launcher(const ptv2_model *m, void *workspace, size_t workspace_bytes, void *stream);
/* g_logits (n0, num_classes); the gradient with respect to `feat` is not formed (the input needs none) */
int ptv2_model_backward_hip_launcher(const ptv2_model *m, const float *g_logits, void *workspace, size_t workspace_bytes,
                                     void *stream);

/* ------------------------------------------------ scene geometry, levels 1.. in one call --
 * Everything of a scene's geometry that lies behind the first grid pooling -- for every stage i: GridPool's clustering of
 * level i (point_transformer_v2m2_base.py:246-268), the self k-NN tables of level i+1 (:223) with their position moments,
 * the 3-NN interpolation table back to level i (:311, libs/pointops/functions/interpolation.py:13-21), and at the end the
 * inverse tables of every table of the scene (level 0's included) -- enqueued by ONE native call instead of ~25 python-level
 * launcher calls (~2 ms of host time per scene, on the critical path of a forward that builds its own geometry).  The sizes
 * of levels 1.. are data dependent: the call reads each pooling's cluster count back (one 4-byte copy + stream
 * synchronisation per stage, on `stream` only) and carves the tables of the next level out of the caller's `arena` as it
 * learns them; `level[].n` and the byte offsets in the struct are its outputs.
 *   in:  num_stages, b, interp, grid_size[], coord0 / offset0 (level 0), level[0].n, level[i].nk + knn[].k (the K values wanted
 *        at every level), knn0[] (level 0's self tables, built by the caller: inputs of the inverse tables)
 *   out: level[i].n (i >= 1) and every `long long` field: byte offset into `arena` of
 *        coord (n,3) f32, offset (b) i32, knn[j].idx (n,k) i32, .mu (3) f64, .cov (9) f64, .inv_ptr (n+1) i32, .inv_rows (n k) i32,
 *        cluster (n) i64, order (n) i32, idx_ptr (n_next + 1) i32, up_idx (n,3) i32, up_w (n,3) f32, up_inv_ptr (n_next + 1),
 *        up_inv_rows (3 n); -1: not produced
 *   sizes_ready / fwd_recorded: progress flags for a caller that runs this call on a helper thread (see the struct).
 *   fwd_ready_event (hipEvent_t, optional): recorded on `stream` once everything the FORWARD needs is enqueued (the inverse
 *        tables, which only the backward reads, follow); knn0_event (optional): `stream` waits for it in front of the inverse
 *        tables (the caller built knn0 on another stream).
 * Returns PTV2_ERR_ARG for a voxel-id overflow (scene extent / grid size beyond the 48-bit sort key). */
#define PTV2_GEO_MAX_K 2
typedef struct ptv2_geo_table {
    int k;
    long long idx, mu, cov, inv_ptr, inv_rows;
} ptv2_geo_table;
typedef struct ptv2_geo_level {
    int n, nk;
    ptv2_geo_table knn[PTV2_GEO_MAX_K];
    long long coord, offset;
    long long cluster, order, idx_ptr, up_idx, up_w, up_inv_ptr, up_inv_rows;
} ptv2_geo_level;
typedef struct ptv2_scene_geo {
    int num_stages, b, interp;
    float grid_size[PTV2_MAX_STAGES];
    const float *coord0;
    const int *offset0;
    const int *knn0[PTV2_GEO_MAX_K];
    void *fwd_ready_event, *knn0_event;
    ptv2_geo_level level[PTV2_MAX_STAGES + 1];
    /* progress, written by the call for a second host thread that polls the struct while the call is still enqueueing:
     * sizes_ready = 1 once the last read-back is in and EVERY n and offset of the struct is final (-1: the call failed);
     * fwd_recorded = 1 once fwd_ready_event has been recorded */
    volatile int sizes_ready, fwd_recorded;
} ptv2_scene_geo;
size_t ptv2_scene_geometry_arena_bytes(const ptv2_scene_geo *g);      /* upper bound, from level[0].n */
size_t ptv2_scene_geometry_workspace_bytes(const ptv2_scene_geo *g);
int ptv2_scene_geometry_hip_launcher(ptv2_scene_geo *g, void *arena, size_t arena_bytes, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* Operand precision of the matrix products launched by the CALLING THREAD through the stand-alone launchers
 * (rows_gemm_*, linear_wgrad_*): bf16 != 0 -> V_MFMA_F32_16X16X32_BF16 on operands rounded to bf16 (fp32 accumulate,
 * fp32 in memory), 0 -> exact fp32 MFMA (default), < 0 -> query only.  Returns the previous setting.  The Block / model
 * launchers take the choice from their `matmul_bf16` field instead. */
int ptv2_matmul_precision(int bf16);

/* sizeof(ptv2_block) [0], sizeof(ptv2_block_grads) [1], sizeof(ptv2_model) [2], sizeof(ptv2_gva_block) [3], sizeof(ptv2_scene_geo) [4]
 * as this library was compiled: bindings
 * that mirror the structs (ctypes) compare it with their own at load time. */
long long ptv2_struct_bytes(int which);

/* ------------------------------------------------ segmentation loss --
 * nn.CrossEntropyLoss(ignore_index) of DefaultSegmentor (pointcept/models/default.py:239-251): mean over the labelled
 * rows of -log softmax(logits)[label]; logits (n,c) fp32, label (n) int64.  loss, count: device scalars; lse (n) is
 * kept for the backward, which writes g_logits = (softmax - onehot) * g_loss / count (0 for ignored rows; all zero
 * when no row is labelled, as torch's CPU kernel).  bad_labels: device scalar, number of labels that are neither
 * ignore_index nor in [0,c) -- torch raises a device assert for those; here the loss becomes nan. */
size_t cross_entropy_workspace_bytes(int n);
int cross_entropy_forward_hip_launcher(int n, int c, const float *logits, const long long *label, int ignore_index,
                                       float *lse, float *loss, float *count, float *bad_labels, void *workspace,
                                       size_t workspace_bytes, void *stream);
int cross_entropy_backward_hip_launcher(int n, int c, const float *logits, const long long *label, int ignore_index,
                                        const float *lse, const float *g_loss, const float *count, float *g_logits,
                                        void *stream);

/* ------------------------------------------------- REAL's logit basket (host memory) --
 * dst (dst_rows, c) fp32 HOST array of one whole scene; dst[ids[r], :] = src[r, :] for r = 0 .. rows-1 in that order:
 * the trainer statement `self.basket[k][ori_idx] = seg` (pointcept/engines/train_sam_real.py:234).  ids int64, src
 * (rows, c) fp32, both host (pinned staging of ao_amd/ptv2/basket.py).  No GPU work, no stream; PTV2_ERR_ARG and nothing
 * written when an id is outside [0, dst_rows). */
int basket_scatter_rows_host(float *dst, long long dst_rows, const long long *ids, const float *src, long long rows, int c);

/* ------------------------------------------------------ optimizer step --
 * torch.optim.AdamW (lr 0.006, weight_decay 0.05 in configs/s3dis/semseg-pt-v2m2-0-base.py:41) over one flat fp32
 * buffer of all parameters: p, g, m (exp_avg), v (exp_avg_sq) of n floats, n % 4 == 0; step counts from 1;
 * grad_scale multiplies g first (1 / world size when g is a sum over ranks). */
int adamw_flat_hip_launcher(long long n, float *p, const float *g, float *m, float *v, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, float grad_scale, void *stream);

/* ------------------------------------------- data-side integer work (§8f) --
 * GridSample (pointcept/datasets/transform.py:769-897): cell (n,3) = floor(coord / grid) computed in fp32,
 * cell_range[0..2] / [3..5] = per-axis min / max of cell, key (n) = FNV64-1A (ravel == 0, transform.py:883-897) or
 * the Fortran-style ravel (ravel != 0, transform.py:865-881) of (cell - min) as uint64.  The caller sorts the keys
 * (unsigned order) and picks one point per run; see ao_amd/ptv2/transform.py. */
int grid_sample_keys_hip_launcher(int n, const float *coord, float grid_x, float grid_y, float grid_z, int ravel,
                                  int *cell, int *cell_range, unsigned long long *key, void *stream);
/* SphereCrop (transform.py:970-981): dist2 (n) = sum((coord - center)^2, 1) with numpy's fp32 rounding sequence
 * ((dx*dx + dy*dy) + dz*dz, products rounded separately); center: 3 floats in device memory. */
int center_dist2_hip_launcher(int n, const float *coord, const float *center, float *dist2, void *stream);
/* Validation counts (pointcept/utils/misc.py:58-70 intersection_and_union_gpu, called from
 * engines/hooks/evaluator.py:136-141 after the k = 1 label transfer :124-134): hist (3,k) int64 = intersection,
 * output and target areas (union = output + target - intersection).  pred (pred_n) int64 class ids; nn_idx (n) int32
 * = nearest predicted point of every target point, or NULL for the identity; target (n) int64.  hist is zeroed here. */
int seg_confusion_hip_launcher(long long n, int k, int ignore_index, const long long *pred, long long pred_n,
                               const int *nn_idx, const long long *target, long long *hist, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PTV2_HIP_H */
