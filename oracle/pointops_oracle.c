/*
 * oracle/pointops_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-thread CPU restatement of the reference's pointops CUDA
 * kernels (jihun1998/AO, libs/pointops/src).  It exists only so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg have something to
 * check the HIP path against.  Nothing under ao_amd/ may import, link or call
 * it.
 *
 * PARITY STATUS: "parity unpinned" against a real CUDA build.  The reference
 * ships no tests, golden vectors or CPU path for these ops (SURVEY.md section
 * 4 / 8c) and its .cu files cannot be built here (no nvcc, no NVIDIA GPU).  The
 * restatement below follows the kernels statement by statement; every function
 * cites the reference lines it follows.
 *
 * Floating point: the reference computes squared distances as
 *     (a-x)*(a-x) + (b-y)*(b-y) + (c-z)*(c-z)
 * and nvcc contracts that with -fmad=true (its default).  We pin the LLVM/NVPTX
 * contraction  fma(dz,dz, fma(dx,dx, dy*dy))  explicitly (REF_D2 below); the
 * HIP kernels use the very same expression, so CPU oracle and GPU agree bit
 * for bit.  Build with -ffp-contract=off so the compiler adds nothing.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float REF_D2(float qx, float qy, float qz, float x, float y, float z) {
    float dx = qx - x, dy = qy - y, dz = qz - z;
    return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
}

/* ------------------------------------------------------------------ kNN -- */
/* libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:15-30 */
static void reheap(float *dist, int *idx, int k) {
    int root = 0;
    int child = root * 2 + 1;
    while (child < k) {
        if (child + 1 < k && dist[child + 1] > dist[child]) child++;
        if (dist[root] > dist[child]) return;
        float td = dist[root]; dist[root] = dist[child]; dist[child] = td;
        int ti = idx[root]; idx[root] = idx[child]; idx[child] = ti;
        root = child;
        child = root * 2 + 1;
    }
}

/* knn_query_cuda_kernel.cu:33-42 */
static void heap_sort(float *dist, int *idx, int k) {
    for (int i = k - 1; i > 0; i--) {
        float td = dist[0]; dist[0] = dist[i]; dist[i] = td;
        int ti = idx[0]; idx[0] = idx[i]; idx[i] = ti;
        reheap(dist, idx, i);
    }
}

/*
 * knn_query_cuda_kernel.cu:60-104 (one CUDA thread == one iteration of the
 * outer loop here).  pad_with_start selects the pointops2 variant
 * (libs/pointops2/src/knnquery/knnquery_cuda_kernel.cu:90: best_idx[i]=start).
 * nsample <= 128 (reference local array size, :82-83).
 */
int oracle_knn_query(int m, int nsample, const float *xyz, const float *new_xyz,
                     const int *offset, const int *new_offset, int *idx, float *dist2,
                     int pad_with_start) {
    if (nsample < 1 || nsample > 128) return -1;
    /* queries are independent (one CUDA thread each); OpenMP, when enabled at
     * build time, only spreads them over host cores for the CPU baseline */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
    for (int pt = 0; pt < m; pt++) {
        float best_dist[128];
        int best_idx[128];
        int bt = 0; /* get_bt_idx (:45-56): first i with pt_idx < new_offset[i] */
        while (!(pt < new_offset[bt])) bt++;
        int start = bt == 0 ? 0 : offset[bt - 1];
        int end = offset[bt];
        float qx = new_xyz[pt * 3 + 0], qy = new_xyz[pt * 3 + 1], qz = new_xyz[pt * 3 + 2];
        for (int i = 0; i < nsample; i++) {
            best_dist[i] = 1e10f;
            best_idx[i] = pad_with_start ? start : -1;
        }
        for (int i = start; i < end; i++) {
            float d2 = REF_D2(qx, qy, qz, xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2]);
            if (d2 < best_dist[0]) {
                best_dist[0] = d2;
                best_idx[0] = i;
                reheap(best_dist, best_idx, nsample);
            }
        }
        heap_sort(best_dist, best_idx, nsample);
        for (int i = 0; i < nsample; i++) {
            idx[(size_t)pt * nsample + i] = best_idx[i];
            dist2[(size_t)pt * nsample + i] = best_dist[i];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ FPS -- */
/* libs/pointops/src/cuda_utils.h:11-14 */
int oracle_opt_n_threads(int work_size) {
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

/*
 * libs/pointops/src/sampling/sampling_cuda_kernel.cu:14-129, simulated thread
 * by thread: `B` = block_size = opt_n_threads(n_max) (launcher :131-134), one
 * block per cloud.  dists/dists_i are the two __shared__ arrays; __update is
 * :5-10 (keeps idx1's entry on ties).  tmp (n floats) must be pre-filled with
 * 1e10 by the caller (libs/pointops/functions/sampling.py:19).
 */
int oracle_farthest_point_sampling(int b, int n_max, const float *xyz, const int *offset,
                                   const int *new_offset, float *tmp, int *idx) {
    const int B = oracle_opt_n_threads(n_max);
    float *dists = (float *)malloc(sizeof(float) * B);
    int *dists_i = (int *)malloc(sizeof(int) * B);
    if (!dists || !dists_i) return -1;
    for (int bid = 0; bid < b; bid++) {
        int start_n, end_n, start_m, end_m, old;
        if (bid == 0) {
            start_n = 0; end_n = offset[0]; start_m = 0; end_m = new_offset[0]; old = 0;
        } else {
            start_n = offset[bid - 1]; end_n = offset[bid];
            start_m = new_offset[bid - 1]; end_m = new_offset[bid];
            old = offset[bid - 1];
        }
        idx[start_m] = start_n; /* :39 */
        for (int j = start_m + 1; j < end_m; j++) {
            float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
            for (int tid = 0; tid < B; tid++) { /* :43-61 */
                int besti = start_n;
                float best = -1.0f;
                for (int k = start_n + tid; k < end_n; k += B) {
                    /* (x2-x1)^2+(y2-y1)^2+(z2-z1)^2 : the point is the minuend here (:54) */
                    float d = REF_D2(xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2], x1, y1, z1);
                    float d2 = fminf(d, tmp[k]);
                    tmp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int s = B / 2; s >= 1; s >>= 1) { /* :63-123 */
                for (int tid = 0; tid < s; tid++) {
                    float v1 = dists[tid], v2 = dists[tid + s];
                    int i1 = dists_i[tid], i2 = dists_i[tid + s];
                    dists[tid] = v1 > v2 ? v1 : v2;
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            }
            old = dists_i[0];
            idx[j] = old;
        }
    }
    free(dists);
    free(dists_i);
    return 0;
}

/* ------------------------------------------------------------- grouping -- */
/* libs/pointops/src/grouping/grouping_cuda_kernel.cu:5-14 (no -1 handling there;
 * here a negative index yields zeros so the oracle never reads out of bounds) */
void oracle_grouping_forward(int m, int nsample, int c, const float *input, const int *idx,
                             float *output) {
    for (size_t r = 0; r < (size_t)m * nsample; r++) {
        int src = idx[r];
        for (int ci = 0; ci < c; ci++)
            output[r * c + ci] = src < 0 ? 0.0f : input[(size_t)src * c + ci];
    }
}

/* grouping_cuda_kernel.cu:16-25; grad_input pre-zeroed by caller (grouping.py:31) */
void oracle_grouping_backward(int m, int nsample, int c, const float *grad_output,
                              const int *idx, float *grad_input) {
    for (size_t r = 0; r < (size_t)m * nsample; r++) {
        int src = idx[r];
        if (src < 0) continue;
        for (int ci = 0; ci < c; ci++) grad_input[(size_t)src * c + ci] += grad_output[r * c + ci];
    }
}

/* -------------------------------------------------------- interpolation -- */
/* libs/pointops/src/interpolation/interpolation_cuda_kernel.cu:5-18; output pre-zeroed */
void oracle_interpolation_forward(int n, int c, int k, const float *input, const int *idx,
                                  const float *weight, float *output) {
    for (int ni = 0; ni < n; ni++)
        for (int ci = 0; ci < c; ci++) {
            float acc = output[(size_t)ni * c + ci];
            for (int i = 0; i < k; i++) {
                size_t ii = (size_t)ni * k + i;
                acc += input[(size_t)idx[ii] * c + ci] * weight[ii];
            }
            output[(size_t)ni * c + ci] = acc;
        }
}

/* interpolation_cuda_kernel.cu:20-33; grad_input pre-zeroed */
void oracle_interpolation_backward(int n, int c, int k, const float *grad_output,
                                   const int *idx, const float *weight, float *grad_input) {
    for (int ni = 0; ni < n; ni++)
        for (int i = 0; i < k; i++) {
            size_t ii = (size_t)ni * k + i;
            for (int ci = 0; ci < c; ci++)
                grad_input[(size_t)idx[ii] * c + ci] += grad_output[(size_t)ni * c + ci] * weight[ii];
        }
}

/* ---------------------------------------------------------- subtraction -- */
/* libs/pointops/src/subtraction/subtraction_cuda_kernel.cu:5-16 */
void oracle_subtraction_forward(int n, int nsample, int c, const float *input1,
                                const float *input2, const int *idx, float *output) {
    for (int ni = 0; ni < n; ni++)
        for (int s = 0; s < nsample; s++) {
            int src = idx[(size_t)ni * nsample + s];
            for (int ci = 0; ci < c; ci++)
                output[((size_t)ni * nsample + s) * c + ci] =
                    input1[(size_t)ni * c + ci] - input2[(size_t)src * c + ci];
        }
}

/* subtraction_cuda_kernel.cu:18-30; both grads pre-zeroed */
void oracle_subtraction_backward(int n, int nsample, int c, const int *idx,
                                 const float *grad_output, float *grad_input1,
                                 float *grad_input2) {
    for (int ni = 0; ni < n; ni++)
        for (int s = 0; s < nsample; s++) {
            int src = idx[(size_t)ni * nsample + s];
            for (int ci = 0; ci < c; ci++) {
                float g = grad_output[((size_t)ni * nsample + s) * c + ci];
                grad_input1[(size_t)ni * c + ci] += g;
                grad_input2[(size_t)src * c + ci] += -g;
            }
        }
}

/* ---------------------------------------------------------- aggregation -- */
/* libs/pointops/src/aggregation/aggregation_cuda_kernel.cu:5-20; output pre-zeroed */
void oracle_aggregation_forward(int n, int nsample, int c, int w_c, const float *input,
                                const float *position, const float *weight, const int *idx,
                                float *output) {
    for (int ni = 0; ni < n; ni++)
        for (int ci = 0; ci < c; ci++) {
            int wi = ci % w_c;
            float acc = output[(size_t)ni * c + ci];
            for (int s = 0; s < nsample; s++) {
                size_t ii = (size_t)ni * nsample + s;
                acc += (input[(size_t)idx[ii] * c + ci] + position[ii * c + ci]) * weight[ii * w_c + wi];
            }
            output[(size_t)ni * c + ci] = acc;
        }
}

/* aggregation_cuda_kernel.cu:22-39; grad_input / grad_weight pre-zeroed */
void oracle_aggregation_backward(int n, int nsample, int c, int w_c, const float *input,
                                 const float *position, const float *weight, const int *idx,
                                 const float *grad_output, float *grad_input,
                                 float *grad_position, float *grad_weight) {
    for (int ni = 0; ni < n; ni++)
        for (int ci = 0; ci < c; ci++) {
            int wi = ci % w_c;
            float go = grad_output[(size_t)ni * c + ci];
            for (int s = 0; s < nsample; s++) {
                size_t ii = (size_t)ni * nsample + s;
                grad_input[(size_t)idx[ii] * c + ci] += go * weight[ii * w_c + wi];
                grad_position[ii * c + ci] = go * weight[ii * w_c + wi];
                grad_weight[ii * w_c + wi] += go * (input[(size_t)idx[ii] * c + ci] + position[ii * c + ci]);
            }
        }
}

/* ------------------------------------------------------------ attention -- */
/* libs/pointops/src/attention/attention_cuda_kernel.cu:9-24; output (m,g) pre-zeroed */
void oracle_attention_relation_step_forward(int m, int g, int c, const float *query,
                                            const float *key, const float *weight,
                                            const int *index_target, const int *index_refer,
                                            float *output) {
    for (int r = 0; r < m; r++)
        for (int gi = 0; gi < g; gi++) {
            float acc = output[(size_t)r * g + gi];
            for (int ci = 0; ci < c; ci++) {
                size_t q = ((size_t)index_target[r] * g + gi) * c + ci;
                size_t kk = ((size_t)index_refer[r] * g + gi) * c + ci;
                acc += query[q] * key[kk] * weight[ci];
            }
            output[(size_t)r * g + gi] = acc;
        }
}

/* attention_cuda_kernel.cu:26-46; grads pre-zeroed */
void oracle_attention_relation_step_backward(int m, int g, int c, const float *query,
                                             float *grad_query, const float *key,
                                             float *grad_key, const float *weight,
                                             float *grad_weight, const int *index_target,
                                             const int *index_refer, const float *grad_output) {
    for (int r = 0; r < m; r++)
        for (int gi = 0; gi < g; gi++) {
            float gr = grad_output[(size_t)r * g + gi];
            for (int ci = 0; ci < c; ci++) {
                size_t q = ((size_t)index_target[r] * g + gi) * c + ci;
                size_t kk = ((size_t)index_refer[r] * g + gi) * c + ci;
                grad_query[q] += gr * key[kk] * weight[ci];
                grad_key[kk] += gr * query[q] * weight[ci];
                grad_weight[ci] += gr * key[kk] * query[q];
            }
        }
}

/* attention_cuda_kernel.cu:49-65; output (n,g,c) pre-zeroed */
void oracle_attention_fusion_step_forward(int m, int g, int c, const float *weight,
                                          const float *value, const int *index_target,
                                          const int *index_refer, float *output) {
    for (int r = 0; r < m; r++)
        for (int gi = 0; gi < g; gi++)
            for (int ci = 0; ci < c; ci++) {
                size_t o = ((size_t)index_target[r] * g + gi) * c + ci;
                size_t v = ((size_t)index_refer[r] * g + gi) * c + ci;
                output[o] += weight[(size_t)r * g + gi] * value[v];
            }
}

/* attention_cuda_kernel.cu:68-86; grads pre-zeroed */
void oracle_attention_fusion_step_backward(int m, int g, int c, const float *weight,
                                           float *grad_weight, const float *value,
                                           float *grad_value, const int *index_target,
                                           const int *index_refer, const float *grad_output) {
    for (int r = 0; r < m; r++)
        for (int gi = 0; gi < g; gi++)
            for (int ci = 0; ci < c; ci++) {
                size_t o = ((size_t)index_target[r] * g + gi) * c + ci;
                size_t v = ((size_t)index_refer[r] * g + gi) * c + ci;
                float go = grad_output[o];
                grad_weight[(size_t)r * g + gi] += go * value[v];
                grad_value[v] += go * weight[(size_t)r * g + gi];
            }
}
