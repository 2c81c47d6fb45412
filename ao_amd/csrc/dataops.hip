// ao_amd/csrc/dataops.hip -- the integer / byte work either side of the training step (gfx950), SURVEY.md §8(f) rows 3-4:
//   * GridSample voxel keys       (pointcept/datasets/transform.py:794-801,865-897)
//   * SphereCrop squared distance (pointcept/datasets/transform.py:970-981)
//   * validation confusion counts (pointcept/utils/misc.py:58-70 fed by engines/hooks/evaluator.py:124-141)
// All HBM-streaming, one lane per point; integer results (keys, counts) are exact, the only floating-point steps
// (coord / grid, squared distance) pin their rounding sequence to numpy's.
#include <limits.h>

#include "gva_common.h"

namespace {

constexpr int DTPB = 256;

__global__ void cell_range_init_kernel(int *range) {
    if (threadIdx.x < 3) range[threadIdx.x] = INT_MAX;
    else if (threadIdx.x < 6) range[threadIdx.x] = INT_MIN;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, WAVE));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, WAVE));
    return v;
}

// cell[i][d] = floor(coord[i][d] / grid[d]) in fp32 (IEEE division, as numpy's float32 true_divide), range[0..2] = per-axis
// minimum, range[3..5] = per-axis maximum (integer atomics: order-independent)
__global__ __launch_bounds__(DTPB) void grid_cells_kernel(int n, const float *__restrict__ coord, float gx, float gy, float gz,
                                                          int *__restrict__ cell, int *range) {
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (long long i = (long long)blockIdx.x * DTPB + threadIdx.x; i < n; i += (long long)gridDim.x * DTPB) {
        const float g[3] = {gx, gy, gz};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int c = (int)floorf(__fdiv_rn(coord[3 * i + d], g[d]));
            cell[3 * i + d] = c;
            lo[d] = min(lo[d], c);
            hi[d] = max(hi[d], c);
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int a = wave_min_i(lo[d]), b = wave_max_i(hi[d]);
        if ((threadIdx.x & 63) == 0) {
            if (a != INT_MAX) atomicMin(range + d, a);
            if (b != INT_MIN) atomicMax(range + 3 + d, b);
        }
    }
}

// FNV64-1A over the three min-shifted cell coordinates (transform.py:883-897) or the Fortran-style ravel
// (transform.py:865-881); both mod 2^64
__global__ __launch_bounds__(DTPB) void grid_keys_kernel(int n, const int *__restrict__ cell, const int *__restrict__ range,
                                                         int ravel, unsigned long long *__restrict__ key) {
    const long long m0 = range[0], m1 = range[1], m2 = range[2];
    const unsigned long long e1 = (unsigned long long)((long long)range[4] - m1 + 1), e2 = (unsigned long long)((long long)range[5] - m2 + 1);
    for (long long i = (long long)blockIdx.x * DTPB + threadIdx.x; i < n; i += (long long)gridDim.x * DTPB) {
        const unsigned long long a = (unsigned long long)((long long)cell[3 * i] - m0), b = (unsigned long long)((long long)cell[3 * i + 1] - m1),
                                 c = (unsigned long long)((long long)cell[3 * i + 2] - m2);
        unsigned long long h;
        if (ravel) {
            h = (a * e1 + b) * e2 + c;
        } else {
            h = 14695981039346656037ULL;
            h *= 1099511628211ULL; h ^= a;
            h *= 1099511628211ULL; h ^= b;
            h *= 1099511628211ULL; h ^= c;
        }
        key[i] = h;
    }
}

// d2[i] = ((dx*dx + dy*dy) + dz*dz) with separately rounded products and sums: numpy's np.sum(np.square(coord - center), 1)
__global__ __launch_bounds__(DTPB) void center_dist2_kernel(int n, const float *__restrict__ coord, const float *__restrict__ center,
                                                            float *__restrict__ d2) {
    const float cx = center[0], cy = center[1], cz = center[2];
    for (long long i = (long long)blockIdx.x * DTPB + threadIdx.x; i < n; i += (long long)gridDim.x * DTPB) {
        const float dx = __fsub_rn(coord[3 * i], cx), dy = __fsub_rn(coord[3 * i + 1], cy), dz = __fsub_rn(coord[3 * i + 2], cz);
        d2[i] = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    }
}

// hist[0][c] = #(pred == target == c), hist[1][c] = #(pred == c, target not ignored), hist[2][c] = #(target == c);
// pred is read through nn (nearest coarse point of every original point) when nn != NULL
__global__ __launch_bounds__(DTPB) void seg_confusion_kernel(long long n, int k, int ignore_index, const long long *__restrict__ pred,
                                                             long long pred_n, const int *__restrict__ nn,
                                                             const long long *__restrict__ target, unsigned long long *hist) {
    extern __shared__ unsigned s_cnt[];  // [3][k]
    for (int e = threadIdx.x; e < 3 * k; e += DTPB) s_cnt[e] = 0u;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * DTPB + threadIdx.x; i < n; i += (long long)gridDim.x * DTPB) {
        const long long t = target[i];
        long long p = ignore_index;
        if (t != ignore_index) {
            const long long j = nn ? (long long)nn[i] : i;
            p = (j >= 0 && j < pred_n) ? pred[j] : (long long)ignore_index;
        }
        if (p >= 0 && p < k) {
            atomicAdd(s_cnt + k + (int)p, 1u);
            if (p == t) atomicAdd(s_cnt + (int)p, 1u);
        }
        if (t >= 0 && t < k) atomicAdd(s_cnt + 2 * k + (int)t, 1u);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * k; e += DTPB)
        if (s_cnt[e]) atomicAdd(hist + e, (unsigned long long)s_cnt[e]);
}

int stream_grid(long long n) { return (int)std::max<long long>(1, std::min<long long>((n + DTPB - 1) / DTPB, 256 * 8)); }

}  // namespace

extern "C" int grid_sample_keys_hip_launcher(int n, const float *coord, float grid_x, float grid_y, float grid_z, int ravel,
                                             int *cell, int *cell_range, unsigned long long *key, void *stream) {
    if (n < 0 || !(grid_x > 0.f) || !(grid_y > 0.f) || !(grid_z > 0.f) || !cell_range) return PTV2_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cell_range_init_kernel, dim3(1), dim3(64), 0, st, cell_range);
    if (n == 0) { PTV2_CHECK_LAUNCH(); return PTV2_OK; }
    if (!coord || !cell || !key) return PTV2_ERR_ARG;
    hipLaunchKernelGGL(grid_cells_kernel, dim3(stream_grid(n)), dim3(DTPB), 0, st, n, coord, grid_x, grid_y, grid_z, cell, cell_range);
    hipLaunchKernelGGL(grid_keys_kernel, dim3(stream_grid(n)), dim3(DTPB), 0, st, n, (const int *)cell, (const int *)cell_range,
                       ravel, key);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int center_dist2_hip_launcher(int n, const float *coord, const float *center, float *dist2, void *stream) {
    if (n < 0 || !center) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    if (!coord || !dist2) return PTV2_ERR_ARG;
    hipLaunchKernelGGL(center_dist2_kernel, dim3(stream_grid(n)), dim3(DTPB), 0, (hipStream_t)stream, n, coord, center, dist2);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int seg_confusion_hip_launcher(long long n, int k, int ignore_index, const long long *pred, long long pred_n,
                                          const int *nn_idx, const long long *target, long long *hist, void *stream) {
    if (n < 0 || k < 1 || k > 4096 || !hist) return PTV2_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(hist, 0, sizeof(long long) * 3 * (size_t)k, st) != hipSuccess) return PTV2_ERR_LAUNCH;
    if (n == 0) return PTV2_OK;
    if (!pred || !target) return PTV2_ERR_ARG;
    hipLaunchKernelGGL(seg_confusion_kernel, dim3(stream_grid(n)), dim3(DTPB), sizeof(unsigned) * 3 * (size_t)k, st, n, k,
                       ignore_index, pred, pred_n, nn_idx, target, (unsigned long long *)hist);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
