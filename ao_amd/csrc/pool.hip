// ao_amd/csrc/pool.hip -- grid-pooling support kernels (GridPool of PT-v2m2,
// pointcept/models/point_transformer_v2/point_transformer_v2m2_base.py:244-269).
//
//   segment_minmax   per-cloud coordinate min / max (torch_scatter.segment_csr(reduce="min"), :249-253),
//                    two launches, no atomics.
//   pool_max fwd     out[j,:] = max over the members of cluster j (segment_csr(reduce="max"), :266) with
//                    the arg-max row recorded (first member wins ties, as a sequential `>` scan does);
//   pool_max bwd     grad_feat[arg[j,c], c] = grad_out[j,c]  -- every (j,c) owns a distinct destination,
//                    so the backward is a plain scatter: no atomics, no N x C mask tensors.
#include <algorithm>

#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MM_CHUNKS = 64;

__global__ __launch_bounds__(TPB) void segment_minmax_partial(int b, const float *__restrict__ xyz,
                                                             const int *__restrict__ offset, float *__restrict__ part) {
    __shared__ float s_lo[TPB / WAVE][3], s_hi[TPB / WAVE][3];
    const int seg = blockIdx.x / MM_CHUNKS, chunk = blockIdx.x - seg * MM_CHUNKS;
    const int start = seg == 0 ? 0 : offset[seg - 1], end = offset[seg];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = start + chunk * TPB + threadIdx.x; i < end; i += MM_CHUNKS * TPB) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3 * (size_t)i + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, WAVE));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, WAVE));
        }
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][a] = lo[a]; s_hi[threadIdx.x >> 6][a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float l = s_lo[0][threadIdx.x], h = s_hi[0][threadIdx.x];
        for (int w = 1; w < TPB / WAVE; ++w) { l = fminf(l, s_lo[w][threadIdx.x]); h = fmaxf(h, s_hi[w][threadIdx.x]); }
        part[(size_t)blockIdx.x * 6 + threadIdx.x] = l;
        part[(size_t)blockIdx.x * 6 + 3 + threadIdx.x] = h;
    }
}

__global__ void segment_minmax_final(int b, const float *__restrict__ part, float *__restrict__ lo, float *__restrict__ hi) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * 3) return;
    const int seg = t / 3, a = t - seg * 3;
    float l = 3.0e38f, h = -3.0e38f;
    for (int ch = 0; ch < MM_CHUNKS; ++ch) {
        l = fminf(l, part[((size_t)seg * MM_CHUNKS + ch) * 6 + a]);
        h = fmaxf(h, part[((size_t)seg * MM_CHUNKS + ch) * 6 + 3 + a]);
    }
    lo[t] = l;
    hi[t] = h;
}

template <int VEC>
__global__ __launch_bounds__(TPB) void pool_max_fwd(long long total, int cv, const float *__restrict__ feat,
                                                    const int *__restrict__ order, const int *__restrict__ idx_ptr,
                                                    float *__restrict__ out, int *__restrict__ arg) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const int j = (int)(e / cv), q = (int)(e - (long long)j * cv);
        const int p0 = idx_ptr[j], p1 = idx_ptr[j + 1];
        if (VEC == 4) {
            int r0 = order[p0];
            float4 best = ((const float4 *)feat)[(size_t)r0 * cv + q];
            int4 bi = make_int4(r0, r0, r0, r0);
            // members 8 at a time: ids, then their rows, then the comparisons in member order (first wins ties as before)
            constexpr int UB = 8;
            for (int p = p0 + 1; p < p1; p += UB) {
                int r[UB];
                float4 v[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) r[u] = p + u < p1 ? order[p + u] : -1;
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (r[u] >= 0) v[u] = ((const float4 *)feat)[(size_t)r[u] * cv + q];
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (r[u] >= 0) {
                        if (v[u].x > best.x) { best.x = v[u].x; bi.x = r[u]; }
                        if (v[u].y > best.y) { best.y = v[u].y; bi.y = r[u]; }
                        if (v[u].z > best.z) { best.z = v[u].z; bi.z = r[u]; }
                        if (v[u].w > best.w) { best.w = v[u].w; bi.w = r[u]; }
                    }
            }
            ((float4 *)out)[e] = best;
            ((int4 *)arg)[e] = bi;
        } else {
            int r0 = order[p0];
            float best = feat[(size_t)r0 * cv + q];
            int bi = r0;
            for (int p = p0 + 1; p < p1; ++p) {
                const int r = order[p];
                const float v = feat[(size_t)r * cv + q];
                if (v > best) { best = v; bi = r; }
            }
            out[e] = best;
            arg[e] = bi;
        }
    }
}

__global__ __launch_bounds__(TPB) void pool_max_bwd(long long total, int c, const float *__restrict__ grad_out,
                                                    const int *__restrict__ arg, float *__restrict__ grad_feat) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const int ch = (int)(e % c);
        grad_feat[(size_t)arg[e] * c + ch] = grad_out[e];
    }
}

}  // namespace

extern "C" size_t segment_minmax_hip_workspace_bytes(int b) { return sizeof(float) * 6 * (size_t)MM_CHUNKS * (b > 0 ? b : 1) + 256; }

extern "C" int segment_minmax_hip_launcher(int b, const float *xyz, const int *offset, float *lo, float *hi,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    if (b < 1) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < segment_minmax_hip_workspace_bytes(b)) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(segment_minmax_partial, dim3(b * MM_CHUNKS), dim3(TPB), 0, st, b, xyz, offset, (float *)workspace);
    hipLaunchKernelGGL(segment_minmax_final, dim3(divup(b * 3, 64)), dim3(64), 0, st, b, (const float *)workspace, lo, hi);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int pool_max_forward_hip_launcher(int n_out, int c, const float *feat, const int *order, const int *idx_ptr,
                                             float *out, int *arg, void *stream) {
    if (n_out < 0 || c < 1) return PTV2_ERR_ARG;
    if (n_out == 0) return PTV2_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = c % 4 == 0 && (((uintptr_t)feat | (uintptr_t)out | (uintptr_t)arg) & 15) == 0;
    const int cv = vec ? c / 4 : c;
    const long long total = (long long)n_out * cv;
    const int nblk = (int)std::min<long long>((total + TPB - 1) / TPB, 256 * 8);
    if (vec) hipLaunchKernelGGL(pool_max_fwd<4>, dim3(nblk), dim3(TPB), 0, st, total, cv, feat, order, idx_ptr, out, arg);
    else hipLaunchKernelGGL(pool_max_fwd<1>, dim3(nblk), dim3(TPB), 0, st, total, cv, feat, order, idx_ptr, out, arg);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int pool_max_backward_hip_launcher(int n_out, int c, const float *grad_out, const int *arg,
                                              float *grad_feat, void *stream) {
    if (n_out < 0 || c < 1) return PTV2_ERR_ARG;
    const long long total = (long long)n_out * c;
    if (total == 0) return PTV2_OK;
    const int nblk = (int)std::min<long long>((total + TPB - 1) / TPB, 256 * 8);
    hipLaunchKernelGGL(pool_max_bwd, dim3(nblk), dim3(TPB), 0, (hipStream_t)stream, total, c, grad_out, arg, grad_feat);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// ------------------------------------------------------------- "map" unpool --
// UnpoolWithSkip backend "map" (point_transformer_v2m2_base.py:305-310): out[i,:] = feat[cluster[i],:]; its gradient
// is the sum over the fine points of each cluster, taken in the CSR order of the pooling (order, idx_ptr) -- a
// fixed-order segment sum instead of the index_put atomics of the stock indexing backward.
namespace {
template <int VEC>
__global__ __launch_bounds__(TPB) void segment_sum_rows(long long total, int cv, const float *__restrict__ grad_fine,
                                                        const int *__restrict__ order, const int *__restrict__ idx_ptr,
                                                        float *__restrict__ grad_coarse) {
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const int j = (int)(e / cv), q = (int)(e - (long long)j * cv);
        if (VEC == 4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p = idx_ptr[j]; p < idx_ptr[j + 1]; ++p) {
                const float4 v = ((const float4 *)grad_fine)[(size_t)order[p] * cv + q];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            ((float4 *)grad_coarse)[e] = acc;
        } else {
            float acc = 0.f;
            for (int p = idx_ptr[j]; p < idx_ptr[j + 1]; ++p) acc += grad_fine[(size_t)order[p] * cv + q];
            grad_coarse[e] = acc;
        }
    }
}
}  // namespace

// grad_coarse (n_out,c) = sum over rows order[idx_ptr[j] .. idx_ptr[j+1]) of grad_fine (.,c)
extern "C" int segment_sum_hip_launcher(int n_out, int c, const float *grad_fine, const int *order, const int *idx_ptr,
                                        float *grad_coarse, void *stream) {
    if (n_out < 0 || c < 1 || !grad_fine || !order || !idx_ptr || !grad_coarse) return PTV2_ERR_ARG;
    if (n_out == 0) return PTV2_OK;
    const bool vec = c % 4 == 0;
    const int cv = vec ? c / 4 : c;
    const long long total = (long long)n_out * cv;
    const int nblk = (int)std::min<long long>((total + TPB - 1) / TPB, 256 * 16);
    if (vec) hipLaunchKernelGGL(segment_sum_rows<4>, dim3(nblk), dim3(TPB), 0, (hipStream_t)stream, total, cv, grad_fine, order, idx_ptr, grad_coarse);
    else hipLaunchKernelGGL(segment_sum_rows<1>, dim3(nblk), dim3(TPB), 0, (hipStream_t)stream, total, cv, grad_fine, order, idx_ptr, grad_coarse);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
