// ao_amd/csrc/gemm.hip -- fp32 row GEMMs of the per-point Linear layers on the CDNA4 matrix cores.
//
// Every nn.Linear of a PT-v2m2 Block (point_transformer_v2m2_base.py:64-66 linear_q/k/v, :143-144 fc1/fc3)
// is Y (m,n) = X (m,k) W^T with m = points (1e3..1e6) and n, k = channels (48..512): 0.5 GFLOP and
// 2 m (n+k) floats of traffic, i.e. HBM-bound by a wide margin.  Library GEMMs spend 40-50 us on the
// m=120k, n=k=48 case whose traffic is worth 8 us, so the block runtime (block.hip) uses this kernel:
//   * V_MFMA_F32_16X16X4_F32 (true fp32, no tf32 rounding), one 16-row strip x BN columns per wavefront,
//     operands swapped (A := W tile, B := X tile) so each lane ends with 4 consecutive output columns of one
//     row and stores a float4;
//   * X / W chunks of 32 reduction indices staged through LDS (row pitch 36 floats: the ds_read_b128 of the
//     16 rows of a strip hit 64 distinct banks), next chunk prefetched into registers during the MFMAs;
//   * W either (n,k) row-major (the forward product) or (k,n) row-major (the input-gradient product
//     gX = gY W reads the same weight matrix with the roles of its two dimensions swapped);
//   * epilogue: + bias[n], optional accumulate onto Y (sums of several products, residual gradients).
#include <algorithm>

#include "common.h"

namespace gemm {

constexpr int BM = 64;        // rows per workgroup (4 wavefronts x 16)
constexpr int KC = 32;        // reduction indices per LDS chunk
constexpr int PITCH = KC + 4; // LDS row pitch in floats
constexpr int THREADS = 256;

typedef float v4f __attribute__((ext_vector_type(4)));

// up to 3 products of one shape in one launch: independent (blockIdx.y selects X/W/bias/Y: the q, k, v projections
// of one input) or summed into one output (the reduction runs over the pairs back to back: g_f1 = sum_i gY_i W_i)
struct GemmMulti {
    const float *X[3], *W[3], *bias[3];
    float *Y[3];
    int count;  // 0: single product from the plain arguments
    int sum;    // != 0: Y[0] = sum_i X[i] op(W[i])
};

template <int BN, bool W_KMAJOR>
__global__ __launch_bounds__(THREADS) void rows_gemm_kernel(int m, int n, int k, const float *__restrict__ X0,
                                                            const float *__restrict__ W0,
                                                            const float *__restrict__ bias0, float *__restrict__ Y0,
                                                            int accumulate, int ncb, GemmMulti multi) {
    __shared__ __attribute__((aligned(16))) float sX[BM * PITCH];
    __shared__ __attribute__((aligned(16))) float sW[BN * PITCH];
    constexpr int NT = BN / 16;            // MFMA column tiles per wavefront
    constexpr int WQ = BN * KC / 4;        // float4 slots of the W chunk
    constexpr int WLOADS = (WQ + THREADS - 1) / THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rb = blockIdx.x / ncb, cb = blockIdx.x - rb * ncb;
    const long long row0 = (long long)rb * BM;
    const int n0 = cb * BN;
    const int z = blockIdx.y;
    const bool indep = multi.count && !multi.sum;
    const float *bias = indep ? multi.bias[z] : bias0;
    float *Y = multi.count ? multi.Y[indep ? z : 0] : Y0;
    const int npair = (multi.count && multi.sum) ? multi.count : 1;
    const float *X = multi.count ? multi.X[indep ? z : 0] : X0;
    const float *W = multi.count ? multi.W[indep ? z : 0] : W0;

    float4 rx[2], rw[WLOADS];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + j * THREADS, r = q >> 3, kq = (q & 7) * 4;
            const long long row = row0 + r;
            rx[j] = (row < m && k0 + kq < k) ? *(const float4 *)(X + row * k + k0 + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < WLOADS; ++j) {
            const int q = tid + j * THREADS;
            rw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < WQ) {
                if (!W_KMAJOR) {
                    const int r = q >> 3, kq = (q & 7) * 4;
                    if (n0 + r < n && k0 + kq < k) rw[j] = *(const float4 *)(W + (long long)(n0 + r) * k + k0 + kq);
                } else {
                    const int kk = q / (BN / 4), cq = (q - kk * (BN / 4)) * 4;
                    if (k0 + kk < k && n0 + cq < n) rw[j] = *(const float4 *)(W + (long long)(k0 + kk) * n + n0 + cq);
                }
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + j * THREADS, r = q >> 3, kq = (q & 7) * 4;
            *(float4 *)(sX + r * PITCH + kq) = rx[j];
        }
#pragma unroll
        for (int j = 0; j < WLOADS; ++j) {
            const int q = tid + j * THREADS;
            if (q < WQ) {
                if (!W_KMAJOR) {
                    const int r = q >> 3, kq = (q & 7) * 4;
                    *(float4 *)(sW + r * PITCH + kq) = rw[j];
                } else {
                    const int kk = q / (BN / 4), cq = (q - kk * (BN / 4)) * 4;
                    sW[(cq + 0) * PITCH + kk] = rw[j].x; sW[(cq + 1) * PITCH + kk] = rw[j].y;
                    sW[(cq + 2) * PITCH + kk] = rw[j].z; sW[(cq + 3) * PITCH + kk] = rw[j].w;
                }
            }
        }
    };

    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};

    // lane (i = lane & 15, s = lane >> 4) owns reduction indices s*8 .. s*8+7 of row / column i in each chunk:
    // the contraction order differs from k-ascending, identically for both operands
    const float *px = sX + (wid * 16 + (lane & 15)) * PITCH + (lane >> 4) * 8;
    const float *pw = sW + (lane & 15) * PITCH + (lane >> 4) * 8;
    for (int pair = 0; pair < npair; ++pair) {
        if (pair > 0) {
            X = multi.X[pair];
            W = multi.W[pair];
            __syncthreads();  // the previous pair's last chunk is still being read
        }
        fetch(0);
        stash();
        __syncthreads();
        for (int k0 = 0; k0 < k; k0 += KC) {
            const bool more = k0 + KC < k;
            if (more) fetch(k0 + KC);
            const float4 x0 = *(const float4 *)px, x1 = *(const float4 *)(px + 4);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 w0 = *(const float4 *)(pw + t * 16 * PITCH), w1 = *(const float4 *)(pw + t * 16 * PITCH + 4);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, x0.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, x0.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, x0.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, x0.w, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, x1.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, x1.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, x1.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, x1.w, acc[t], 0, 0, 0);
            }
            if (more) {
                __syncthreads();
                stash();
                __syncthreads();
            }
        }
    }
    // D[i][j]: i = output column within the tile = (lane >> 4) * 4 + reg, j = row within the strip = lane & 15
    const long long row = row0 + wid * 16 + (lane & 15);
    if (row < m) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = n0 + t * 16 + (lane >> 4) * 4;
            if (col < n) {
                float4 v = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                if (bias) {
                    const float4 bb = *(const float4 *)(bias + col);
                    v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                }
                float4 *dst = (float4 *)(Y + row * n + col);
                if (accumulate) {
                    const float4 o = *dst;
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *dst = v;
            }
        }
    }
}

}  // namespace gemm

// Y (m,n) [+]= X (m,k) op(W) + bias.  w_kmajor == 0: W is (n,k) row-major (y = x W^T, nn.Linear forward);
// w_kmajor != 0: W is (k,n) row-major (gx = gy W).  n % 4 == 0, k % 4 == 0; bias may be NULL.
extern "C" int rows_gemm_hip_launcher(int m, int n, int k, const float *X, const float *W, int w_kmajor,
                                      const float *bias, float *Y, int accumulate, void *stream) {
    using namespace gemm;
    if (m < 0 || n < 4 || k < 4 || n % 4 != 0 || k % 4 != 0 || !X || !W || !Y) return PTV2_ERR_ARG;
    if (m == 0) return PTV2_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool n48 = n % 48 == 0;  // 48 / 96 / 192 / 384: column blocks of 48 waste nothing
    const int bn = n48 ? 48 : 64;
    const int ncb = (n + bn - 1) / bn;
    const long long nrb = ((long long)m + BM - 1) / BM;
    if (nrb * ncb > 2147483647LL) return PTV2_ERR_ARG;
    const dim3 grid((unsigned)(nrb * ncb));
    {
        PtvScopedTimer t(KID_ROWS_GEMM, st, 4.0 * ((double)m * (n + k) + (double)n * k));
        if (n48) {
            if (w_kmajor) hipLaunchKernelGGL((rows_gemm_kernel<48, true>), grid, dim3(THREADS), 0, st, m, n, k, X, W, bias, Y, accumulate, ncb, GemmMulti{});
            else hipLaunchKernelGGL((rows_gemm_kernel<48, false>), grid, dim3(THREADS), 0, st, m, n, k, X, W, bias, Y, accumulate, ncb, GemmMulti{});
        } else {
            if (w_kmajor) hipLaunchKernelGGL((rows_gemm_kernel<64, true>), grid, dim3(THREADS), 0, st, m, n, k, X, W, bias, Y, accumulate, ncb, GemmMulti{});
            else hipLaunchKernelGGL((rows_gemm_kernel<64, false>), grid, dim3(THREADS), 0, st, m, n, k, X, W, bias, Y, accumulate, ncb, GemmMulti{});
        }
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

// count (<= 3) products of one shape in one launch.  sum == 0: Y[i] = X[i] op(W[i]) + bias[i] (independent outputs);
// sum != 0: Y[0] (+)= sum_i X[i] op(W[i]) (+ bias[0]).
extern "C" int rows_gemm_multi_hip_launcher(int m, int n, int k, int count, int sum, const float *const *X,
                                            const float *const *W, int w_kmajor, const float *const *bias, float *const *Y,
                                            int accumulate, void *stream) {
    using namespace gemm;
    if (m < 0 || n < 4 || k < 4 || n % 4 != 0 || k % 4 != 0 || count < 1 || count > 3 || !X || !W || !Y) return PTV2_ERR_ARG;
    if (m == 0) return PTV2_OK;
    GemmMulti gm{};
    gm.count = count;
    gm.sum = sum ? 1 : 0;
    for (int i = 0; i < count; ++i) {
        if (!X[i] || !W[i] || (!Y[i] && (i == 0 || !sum))) return PTV2_ERR_ARG;
        gm.X[i] = X[i]; gm.W[i] = W[i]; gm.bias[i] = bias ? bias[i] : nullptr; gm.Y[i] = Y[i];
    }
    hipStream_t st = (hipStream_t)stream;
    const bool n48 = n % 48 == 0;
    const int bn = n48 ? 48 : 64;
    const int ncb = (n + bn - 1) / bn;
    const long long nrb = ((long long)m + BM - 1) / BM;
    if (nrb * ncb > 2147483647LL) return PTV2_ERR_ARG;
    const dim3 grid((unsigned)(nrb * ncb), sum ? 1 : count);
    const float *b0 = sum ? gm.bias[0] : nullptr;
    {
        PtvScopedTimer t(KID_ROWS_GEMM, st, 4.0 * count * ((double)m * (n + k) + (double)n * k));
        if (n48) {
            if (w_kmajor) hipLaunchKernelGGL((rows_gemm_kernel<48, true>), grid, dim3(THREADS), 0, st, m, n, k, gm.X[0], gm.W[0], b0, gm.Y[0], accumulate, ncb, gm);
            else hipLaunchKernelGGL((rows_gemm_kernel<48, false>), grid, dim3(THREADS), 0, st, m, n, k, gm.X[0], gm.W[0], b0, gm.Y[0], accumulate, ncb, gm);
        } else {
            if (w_kmajor) hipLaunchKernelGGL((rows_gemm_kernel<64, true>), grid, dim3(THREADS), 0, st, m, n, k, gm.X[0], gm.W[0], b0, gm.Y[0], accumulate, ncb, gm);
            else hipLaunchKernelGGL((rows_gemm_kernel<64, false>), grid, dim3(THREADS), 0, st, m, n, k, gm.X[0], gm.W[0], b0, gm.Y[0], accumulate, ncb, gm);
        }
    }
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
