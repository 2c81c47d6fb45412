// ao_amd/csrc/gather_ops.hip -- the index-gather family of the pointops API on gfx950:
// grouping, interpolation, subtraction, aggregation, attention relation / fusion steps.
//
// Semantics follow libs/pointops/src/{grouping,interpolation,subtraction,aggregation,attention}
// (file:line per kernel below).  The thread mappings are not the reference's: rows are moved as
// float4 where the channel count allows it (one lane = 16 B, a wavefront = 1 KiB, guide G13),
// per-output reductions run over the neighbour axis inside one thread instead of through
// atomics wherever the output row is owned by one thread, and launches are grid-stride with
// at most 8 workgroups per CU.  Scatter-adds whose destination is data dependent keep fp32
// atomics like the reference (sum order is unspecified there too).
#include "common.h"

namespace {

constexpr int TPB = 256;
inline int grid_for(long long work) { return (int)std::min<long long>(divup(work, TPB), 256 * 8); }

// ------------------------------------------------------------------ grouping --
// grouping_cuda_kernel.cu:5-14.  VEC = 4 when c % 4 == 0.
template <int VEC>
__global__ __launch_bounds__(TPB) void grouping_fwd(long long total, int cv, const float *__restrict__ input,
                                                    const int *__restrict__ idx, float *__restrict__ output) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long row = e / cv;
        int ci = (int)(e - row * cv);
        int src = idx[row];
        if (VEC == 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src >= 0) v = ((const float4 *)input)[(long long)src * cv + ci];
            ((float4 *)output)[e] = v;
        } else {
            output[e] = src >= 0 ? input[(long long)src * cv + ci] : 0.f;
        }
    }
}

// grouping_cuda_kernel.cu:16-25
__global__ __launch_bounds__(TPB) void grouping_bwd(long long total, int c, const float *__restrict__ grad_output,
                                                    const int *__restrict__ idx, float *grad_input) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long row = e / c;
        int ci = (int)(e - row * c);
        int src = idx[row];
        if (src >= 0) atomicAdd(grad_input + (long long)src * c + ci, grad_output[e]);
    }
}

// ------------------------------------------------------------- interpolation --
// interpolation_cuda_kernel.cu:5-18 (accumulates in the reference's i = 0..k-1 order onto the
// caller-zeroed output)
__global__ __launch_bounds__(TPB) void interpolation_fwd(long long total, int c, int k,
                                                         const float *__restrict__ input,
                                                         const int *__restrict__ idx,
                                                         const float *__restrict__ weight, float *output) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long ni = e / c;
        int ci = (int)(e - ni * c);
        float acc = output[e];
        for (int i = 0; i < k; ++i) {
            int src = idx[ni * k + i];
            acc += input[(long long)src * c + ci] * weight[ni * k + i];
        }
        output[e] = acc;
    }
}

// interpolation_cuda_kernel.cu:20-33
__global__ __launch_bounds__(TPB) void interpolation_bwd(long long total, int c, int k,
                                                         const float *__restrict__ grad_output,
                                                         const int *__restrict__ idx,
                                                         const float *__restrict__ weight, float *grad_input) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long ni = e / c;
        int ci = (int)(e - ni * c);
        float g = grad_output[e];
        for (int i = 0; i < k; ++i) {
            int src = idx[ni * k + i];
            atomicAdd(grad_input + (long long)src * c + ci, g * weight[ni * k + i]);
        }
    }
}

// the same gradient as a fixed-order gather through the inverse table of idx (inv_ptr (>= m+1), inv_rows: the slots
// r = n*k + i with idx[r] == j, ascending): no float atomics, bitwise reproducible, no zero-fill of grad_input
template <int VEC>
__global__ __launch_bounds__(TPB) void interpolation_bwd_gather(long long total, int cv, int k,
                                                                const float *__restrict__ grad_output,
                                                                const int *__restrict__ inv_ptr,
                                                                const int *__restrict__ inv_rows,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ grad_input) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const long long j = e / cv;
        const int q = (int)(e - j * cv);
        float acc[VEC];
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[t] = 0.f;
        // the list is walked 8 entries at a time: slot ids, then all weights and gradient rows, then the sums in list order
        // (entry by entry a lane paid two dependent memory latencies per entry, ~19 entries per coarse point)
        constexpr int UB = 8;
        const int p0 = inv_ptr[j], p1 = inv_ptr[j + 1];
        for (int p = p0; p < p1; p += UB) {
            int r[UB];
            float w[UB], t[UB][VEC];
#pragma unroll
            for (int u = 0; u < UB; ++u) r[u] = p + u < p1 ? inv_rows[p + u] : -1;
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (r[u] >= 0) {
                    w[u] = weight[r[u]];
                    const float *go = grad_output + ((long long)(r[u] / k) * cv + q) * VEC;
                    if (VEC == 4) {
                        const float4 t4 = *(const float4 *)go;
                        t[u][0] = t4.x; t[u][1 % VEC] = t4.y; t[u][2 % VEC] = t4.z; t[u][3 % VEC] = t4.w;
                    } else {
                        t[u][0] = go[0];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (r[u] >= 0) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] = __builtin_fmaf(w[u], t[u][i], acc[i]);
                }
        }
        float *dst = grad_input + (j * cv + q) * VEC;
        if (VEC == 4) *(float4 *)dst = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else dst[0] = acc[0];
    }
}

// --------------------------------------------------------------- subtraction --
// subtraction_cuda_kernel.cu:5-16
template <int VEC>
__global__ __launch_bounds__(TPB) void subtraction_fwd(long long total, int nsample, int cv,
                                                       const float *__restrict__ input1,
                                                       const float *__restrict__ input2,
                                                       const int *__restrict__ idx, float *__restrict__ output) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long row = e / cv;
        int ci = (int)(e - row * cv);
        long long ni = row / nsample;
        int src = idx[row];
        if (VEC == 4) {
            float4 a = ((const float4 *)input1)[ni * cv + ci];
            float4 b = ((const float4 *)input2)[(long long)src * cv + ci];
            ((float4 *)output)[e] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        } else {
            output[e] = input1[ni * cv + ci] - input2[(long long)src * cv + ci];
        }
    }
}

// subtraction_cuda_kernel.cu:18-30.  grad_input1[n,c] is owned by one thread (sum over the
// neighbour axis in s order, no atomics); grad_input2 is a data-dependent scatter (atomics).
__global__ __launch_bounds__(TPB) void subtraction_bwd(long long total, int nsample, int c,
                                                       const int *__restrict__ idx,
                                                       const float *__restrict__ grad_output, float *grad_input1,
                                                       float *grad_input2) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long ni = e / c;
        int ci = (int)(e - ni * c);
        float acc = grad_input1[e];
        for (int s = 0; s < nsample; ++s) {
            float g = grad_output[(ni * nsample + s) * c + ci];
            acc += g;
            atomicAdd(grad_input2 + (long long)idx[ni * nsample + s] * c + ci, -g);
        }
        grad_input1[e] = acc;
    }
}

// --------------------------------------------------------------- aggregation --
// aggregation_cuda_kernel.cu:5-20
__global__ __launch_bounds__(TPB) void aggregation_fwd(long long total, int nsample, int c, int w_c,
                                                       const float *__restrict__ input,
                                                       const float *__restrict__ position,
                                                       const float *__restrict__ weight,
                                                       const int *__restrict__ idx, float *output) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long ni = e / c;
        int ci = (int)(e - ni * c);
        int wi = ci % w_c;
        float acc = output[e];
        for (int s = 0; s < nsample; ++s) {
            long long ii = ni * nsample + s;
            acc += (input[(long long)idx[ii] * c + ci] + position[ii * c + ci]) * weight[ii * w_c + wi];
        }
        output[e] = acc;
    }
}

// aggregation_cuda_kernel.cu:22-39
__global__ __launch_bounds__(TPB) void aggregation_bwd(long long total, int nsample, int c, int w_c,
                                                       const float *__restrict__ input,
                                                       const float *__restrict__ position,
                                                       const float *__restrict__ weight,
                                                       const int *__restrict__ idx,
                                                       const float *__restrict__ grad_output, float *grad_input,
                                                       float *__restrict__ grad_position, float *grad_weight) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long ni = e / c;
        int ci = (int)(e - ni * c);
        int wi = ci % w_c;
        float go = grad_output[e];
        for (int s = 0; s < nsample; ++s) {
            long long ii = ni * nsample + s;
            long long src = (long long)idx[ii] * c + ci;
            float w = weight[ii * w_c + wi];
            atomicAdd(grad_input + src, go * w);
            grad_position[ii * c + ci] = go * w;
            atomicAdd(grad_weight + ii * w_c + wi, go * (input[src] + position[ii * c + ci]));
        }
    }
}

// ----------------------------------------------------------------- attention --
// attention_cuda_kernel.cu:9-24: one thread per (r, g) owns the output element, c summed in order.
__global__ __launch_bounds__(TPB) void attn_relation_fwd(long long total, int g, int c,
                                                         const float *__restrict__ query,
                                                         const float *__restrict__ key,
                                                         const float *__restrict__ weight,
                                                         const int *__restrict__ tgt, const int *__restrict__ ref,
                                                         float *output) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long r = e / g;
        int gi = (int)(e - r * g);
        const float *q = query + ((long long)tgt[r] * g + gi) * c;
        const float *k = key + ((long long)ref[r] * g + gi) * c;
        float acc = output[e];
        for (int ci = 0; ci < c; ++ci) acc += q[ci] * k[ci] * weight[ci];
        output[e] = acc;
    }
}

// attention_cuda_kernel.cu:26-46
__global__ __launch_bounds__(TPB) void attn_relation_bwd(long long total, int g, int c,
                                                         const float *__restrict__ query, float *grad_query,
                                                         const float *__restrict__ key, float *grad_key,
                                                         const float *__restrict__ weight, float *grad_weight,
                                                         const int *__restrict__ tgt, const int *__restrict__ ref,
                                                         const float *__restrict__ grad_output) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long rg = e / c;
        int ci = (int)(e - rg * c);
        long long r = rg / g;
        int gi = (int)(rg - r * g);
        long long qi = ((long long)tgt[r] * g + gi) * c + ci;
        long long ki = ((long long)ref[r] * g + gi) * c + ci;
        float gr = grad_output[rg];
        float qv = query[qi], kv = key[ki], wv = weight[ci];
        atomicAdd(grad_query + qi, gr * kv * wv);
        atomicAdd(grad_key + ki, gr * qv * wv);
        atomicAdd(grad_weight + ci, gr * kv * qv);
    }
}

// attention_cuda_kernel.cu:49-65
__global__ __launch_bounds__(TPB) void attn_fusion_fwd(long long total, int g, int c,
                                                       const float *__restrict__ weight,
                                                       const float *__restrict__ value,
                                                       const int *__restrict__ tgt, const int *__restrict__ ref,
                                                       float *output) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long rg = e / c;
        int ci = (int)(e - rg * c);
        long long r = rg / g;
        int gi = (int)(rg - r * g);
        float f = weight[rg] * value[((long long)ref[r] * g + gi) * c + ci];
        atomicAdd(output + ((long long)tgt[r] * g + gi) * c + ci, f);
    }
}

// attention_cuda_kernel.cu:68-86: grad_weight[r,g] is owned by one thread (c summed in order).
__global__ __launch_bounds__(TPB) void attn_fusion_bwd(long long total, int g, int c,
                                                       const float *__restrict__ weight, float *grad_weight,
                                                       const float *__restrict__ value, float *grad_value,
                                                       const int *__restrict__ tgt, const int *__restrict__ ref,
                                                       const float *__restrict__ grad_output) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        long long r = e / g;
        int gi = (int)(e - r * g);
        long long oi = ((long long)tgt[r] * g + gi) * c;
        long long vi = ((long long)ref[r] * g + gi) * c;
        float w = weight[e];
        float acc = grad_weight[e];
        for (int ci = 0; ci < c; ++ci) {
            float go = grad_output[oi + ci];
            acc += go * value[vi + ci];
            atomicAdd(grad_value + vi + ci, go * w);
        }
        grad_weight[e] = acc;
    }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int grouping_forward_hip_launcher(int m, int nsample, int c, const float *input, const int *idx,
                                             float *output, void *stream) {
    if (m < 0 || nsample < 0 || c < 0) return PTV2_ERR_ARG;
    long long rows = (long long)m * nsample;
    if (rows == 0 || c == 0) return PTV2_OK;
    if (c % 4 == 0 && aligned16(input) && aligned16(output)) {
        long long total = rows * (c / 4);
        hipLaunchKernelGGL(grouping_fwd<4>, dim3(grid_for(total)), dim3(TPB), 0, ST, total, c / 4, input, idx, output);
    } else {
        long long total = rows * c;
        hipLaunchKernelGGL(grouping_fwd<1>, dim3(grid_for(total)), dim3(TPB), 0, ST, total, c, input, idx, output);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int grouping_backward_hip_launcher(int m, int nsample, int c, const float *grad_output,
                                              const int *idx, float *grad_input, void *stream) {
    if (m < 0 || nsample < 0 || c < 0) return PTV2_ERR_ARG;
    long long total = (long long)m * nsample * c;
    if (total == 0) return PTV2_OK;
    hipLaunchKernelGGL(grouping_bwd, dim3(grid_for(total)), dim3(TPB), 0, ST, total, c, grad_output, idx, grad_input);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// weights of the inverse-distance interpolation from the k-NN's squared distances, and the reference's wrap of a missing
// neighbour: w_s = (1 / (sqrt(d2_s) + 1e-8)) / sum_t (1 / (sqrt(d2_t) + 1e-8)) (libs/pointops/functions/interpolation.py:
// 13-16, sums in slot order); idx < 0 -> idx + n (torch's negative indexing, :21).  One launch for what the python
// statement spends nine on.
__global__ __launch_bounds__(256) void interpolation_weights_kernel(long long m, int k, int n, const float *__restrict__ dist2,
                                                                    int *__restrict__ idx, float *__restrict__ weight) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long long)gridDim.x * 256) {
        float r[8];
        float norm = 0.f;
        for (int s = 0; s < k; ++s) {
            r[s & 7] = __fdiv_rn(1.0f, __fadd_rn(__fsqrt_rn(dist2[i * k + s]), 1e-8f));
            norm = __fadd_rn(norm, r[s & 7]);
        }
        for (int s = 0; s < k; ++s) {
            weight[i * k + s] = __fdiv_rn(r[s & 7], norm);
            const int j = idx[i * k + s];
            if (j < 0) idx[i * k + s] = j + n;
        }
    }
}

extern "C" int interpolation_weights_hip_launcher(int m, int k, int n, const float *dist2, int *idx, float *weight, void *stream) {
    if (m < 0 || k < 1 || k > 8 || n < 0) return PTV2_ERR_ARG;
    if (m == 0) return PTV2_OK;
    if (!dist2 || !idx || !weight) return PTV2_ERR_ARG;
    const int nblk = (int)std::min<long long>(((long long)m + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(interpolation_weights_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (long long)m, k, n, dist2, idx,
                       weight);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int interpolation_forward_hip_launcher(int n, int c, int k, const float *input, const int *idx,
                                                  const float *weight, float *output, void *stream) {
    if (n < 0 || c < 0 || k < 0) return PTV2_ERR_ARG;
    long long total = (long long)n * c;
    if (total == 0) return PTV2_OK;
    hipLaunchKernelGGL(interpolation_fwd, dim3(grid_for(total)), dim3(TPB), 0, ST, total, c, k, input, idx, weight,
                       output);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int interpolation_backward_hip_launcher(int n, int c, int k, const float *grad_output,
                                                   const int *idx, const float *weight, float *grad_input,
                                                   void *stream) {
    if (n < 0 || c < 0 || k < 0) return PTV2_ERR_ARG;
    long long total = (long long)n * c;
    if (total == 0) return PTV2_OK;
    hipLaunchKernelGGL(interpolation_bwd, dim3(grid_for(total)), dim3(TPB), 0, ST, total, c, k, grad_output, idx,
                       weight, grad_input);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// grad_input (m,c) = scatter of grad_output (n,c) through idx (n,k), as a gather over the inverse table of idx
// (inverse_table_hip_launcher); every row of grad_input is written (rows nobody points at get zeros)
extern "C" int interpolation_backward_gather_hip_launcher(int m, int c, int k, const float *grad_output, const int *inv_ptr,
                                                          const int *inv_rows, const float *weight, float *grad_input,
                                                          void *stream) {
    if (m < 0 || c < 0 || k < 1 || !inv_ptr || !inv_rows) return PTV2_ERR_ARG;
    if ((long long)m * c == 0) return PTV2_OK;
    if (c % 4 == 0 && aligned16(grad_output) && aligned16(grad_input)) {
        const long long total = (long long)m * (c / 4);
        hipLaunchKernelGGL(interpolation_bwd_gather<4>, dim3(grid_for(total)), dim3(TPB), 0, ST, total, c / 4, k, grad_output,
                           inv_ptr, inv_rows, weight, grad_input);
    } else {
        const long long total = (long long)m * c;
        hipLaunchKernelGGL(interpolation_bwd_gather<1>, dim3(grid_for(total)), dim3(TPB), 0, ST, total, c, k, grad_output, inv_ptr,
                           inv_rows, weight, grad_input);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int subtraction_forward_hip_launcher(int n, int nsample, int c, const float *input1,
                                                const float *input2, const int *idx, float *output,
                                                void *stream) {
    if (n < 0 || nsample < 0 || c < 0) return PTV2_ERR_ARG;
    long long rows = (long long)n * nsample;
    if (rows == 0 || c == 0) return PTV2_OK;
    if (c % 4 == 0 && aligned16(input1) && aligned16(input2) && aligned16(output)) {
        long long total = rows * (c / 4);
        hipLaunchKernelGGL(subtraction_fwd<4>, dim3(grid_for(total)), dim3(TPB), 0, ST, total, nsample, c / 4, input1,
                           input2, idx, output);
    } else {
        long long total = rows * c;
        hipLaunchKernelGGL(subtraction_fwd<1>, dim3(grid_for(total)), dim3(TPB), 0, ST, total, nsample, c, input1,
                           input2, idx, output);
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int subtraction_backward_hip_launcher(int n, int nsample, int c, const int *idx,
                                                 const float *grad_output, float *grad_input1,
                                                 float *grad_input2, void *stream) {
    if (n < 0 || nsample < 0 || c < 0) return PTV2_ERR_ARG;
    long long total = (long long)n * c;
    if (total == 0) return PTV2_OK;
    hipLaunchKernelGGL(subtraction_bwd, dim3(grid_for(total)), dim3(TPB), 0, ST, total, nsample, c, idx, grad_output,
                       grad_input1, grad_input2);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int aggregation_forward_hip_launcher(int n, int nsample, int c, int w_c, const float *input,
                                                const float *position, const float *weight, const int *idx,
                                                float *output, void *stream) {
    if (n < 0 || nsample < 0 || c < 0 || w_c < 1) return PTV2_ERR_ARG;
    long long total = (long long)n * c;
    if (total == 0) return PTV2_OK;
    hipLaunchKernelGGL(aggregation_fwd, dim3(grid_for(total)), dim3(TPB), 0, ST, total, nsample, c, w_c, input,
                       position, weight, idx, output);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int aggregation_backward_hip_launcher(int n, int nsample, int c, int w_c, const float *input,
                                                 const float *position, const float *weight, const int *idx,
                                                 const float *grad_output, float *grad_input,
                                                 float *grad_position, float *grad_weight, void *stream) {
    if (n < 0 || nsample < 0 || c < 0 || w_c < 1) return PTV2_ERR_ARG;
    long long total = (long long)n * c;
    if (total == 0) return PTV2_OK;
    hipLaunchKernelGGL(aggregation_bwd, dim3(grid_for(total)), dim3(TPB), 0, ST, total, nsample, c, w_c, input,
                       position, weight, idx, grad_output, grad_input, grad_position, grad_weight);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int attention_relation_step_forward_hip_launcher(int m, int g, int c, const float *query,
                                                            const float *key, const float *weight,
                                                            const int *index_target, const int *index_refer,
                                                            float *output, void *stream) {
    if (m < 0 || g < 0 || c < 0) return PTV2_ERR_ARG;
    long long total = (long long)m * g;
    if (total == 0) return PTV2_OK;
    hipLaunchKernelGGL(attn_relation_fwd, dim3(grid_for(total)), dim3(TPB), 0, ST, total, g, c, query, key, weight,
                       index_target, index_refer, output);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int attention_relation_step_backward_hip_launcher(int m, int g, int c, const float *query,
                                                             float *grad_query, const float *key,
                                                             float *grad_key, const float *weight,
                                                             float *grad_weight, const int *index_target,
                                                             const int *index_refer, const float *grad_output,
                                                             void *stream) {
    if (m < 0 || g < 0 || c < 0) return PTV2_ERR_ARG;
    long long total = (long long)m * g * c;
    if (total == 0) return PTV2_OK;
    hipLaunchKernelGGL(attn_relation_bwd, dim3(grid_for(total)), dim3(TPB), 0, ST, total, g, c, query, grad_query, key,
                       grad_key, weight, grad_weight, index_target, index_refer, grad_output);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int attention_fusion_step_forward_hip_launcher(int m, int g, int c, const float *weight,
                                                          const float *value, const int *index_target,
                                                          const int *index_refer, float *output, void *stream) {
    if (m < 0 || g < 0 || c < 0) return PTV2_ERR_ARG;
    long long total = (long long)m * g * c;
    if (total == 0) return PTV2_OK;
    hipLaunchKernelGGL(attn_fusion_fwd, dim3(grid_for(total)), dim3(TPB), 0, ST, total, g, c, weight, value,
                       index_target, index_refer, output);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int attention_fusion_step_backward_hip_launcher(int m, int g, int c, const float *weight,
                                                           float *grad_weight, const float *value,
                                                           float *grad_value, const int *index_target,
                                                           const int *index_refer, const float *grad_output,
                                                           void *stream) {
    if (m < 0 || g < 0 || c < 0) return PTV2_ERR_ARG;
    long long total = (long long)m * g;
    if (total == 0) return PTV2_OK;
    hipLaunchKernelGGL(attn_fusion_bwd, dim3(grid_for(total)), dim3(TPB), 0, ST, total, g, c, weight, grad_weight,
                       value, grad_value, index_target, index_refer, grad_output);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
