// tools/probes/rows_gemm_stream_probe.hip -- Y (m,48) = X (m,48) W^T (48,48), exact-fp32 MFMA, m = 120 000: does a
// persistent, software-pipelined form stream faster than one 64-row block per workgroup?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/rows_gemm_stream_probe.hip -o /tmp/rgs && /tmp/rgs
//   oneshot  one 64-row block per 256-thread workgroup, W staged through LDS behind a barrier (the shape of
//            gemm.hip's rows_gemm_direct_kernel<48, false, 48>, without its epilogue records)
//   stream   W fragments in registers for the whole launch (36 VGPRs), a wavefront walks 16-row strips with the next
//            strip's rows requested before the current one is multiplied; no LDS, no barrier; every store unconditional
// Both read 23 MB and write 23 MB; the elementwise kernels of the step move such bytes at 5.6 TB/s (8.2 us).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int K = 48, N = 48, NT = 3, QF = 3, LDW = K + 8;

__global__ __launch_bounds__(256) void oneshot_kernel(int m, const float *__restrict__ X, const float *__restrict__ W,
                                                      float *__restrict__ Y) {
    __shared__ __attribute__((aligned(16))) float sW[N * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const long long row = (long long)blockIdx.x * 64 + wid * 16 + l15;
    const long long rr = row < m ? row : m - 1;
    float4 x[QF];
#pragma unroll
    for (int j = 0; j < QF; ++j) x[j] = *(const float4 *)(X + rr * K + 16 * j + 4 * q);
    for (int e = tid; e < N * (K / 4); e += 256) {
        const int r = e / (K / 4), k4 = e - r * (K / 4);
        *(float4 *)(sW + r * LDW + 4 * k4) = *(const float4 *)(W + r * K + 4 * k4);
    }
    __syncthreads();
    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < QF; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 w4 = *(const float4 *)(sW + (16 * t + l15) * LDW + 16 * j + 4 * q);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, x[j].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, x[j].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, x[j].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, x[j].w, acc[t], 0, 0, 0);
        }
    if (row < m) {
#pragma unroll
        for (int t = 0; t < NT; ++t) *(float4 *)(Y + row * N + 16 * t + 4 * q) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
}

template <int AHEAD>
__global__ __launch_bounds__(256) void stream_kernel(int m, const float *__restrict__ X, const float *__restrict__ W,
                                                     float *__restrict__ Y) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q = lane >> 4;
    float4 w[NT][QF];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < QF; ++j) w[t][j] = *(const float4 *)(W + (16 * t + l15) * K + 16 * j + 4 * q);
    const int strips = (m + 15) / 16;
    const int stride = gridDim.x * 4;
    const unsigned last = (unsigned)(m - 1);
    auto load = [&](int s, float4 (&x)[QF]) {
        unsigned row = (unsigned)s * 16u + l15;
        row = row < (unsigned)m ? row : last;
        const char *p = (const char *)X + row * (4u * K) + 16u * q;
#pragma unroll
        for (int j = 0; j < QF; ++j) x[j] = *(const float4 *)(p + 64 * j);
    };
    float4 xn[AHEAD][QF];
    int s = blockIdx.x * 4 + wid;
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) load(s + a * stride, xn[a]);
    __builtin_amdgcn_s_waitcnt(0);
    for (; s < strips; s += stride) {
        float4 x[QF];
#pragma unroll
        for (int j = 0; j < QF; ++j) x[j] = xn[0][j];
#pragma unroll
        for (int a = 0; a + 1 < AHEAD; ++a)
#pragma unroll
            for (int j = 0; j < QF; ++j) xn[a][j] = xn[a + 1][j];
        load(s + AHEAD * stride, xn[AHEAD - 1]);
        v4f acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < QF; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][j].x, x[j].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][j].y, x[j].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][j].z, x[j].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][j].w, x[j].w, acc[t], 0, 0, 0);
            }
        unsigned row = (unsigned)s * 16u + l15;
        row = row < (unsigned)m ? row : last;  // (a row past the end was loaded as the last row: the same values again)
        char *yp = (char *)Y + row * (4u * N) + 16u * q;
#pragma unroll
        for (int t = 0; t < NT; ++t) *(float4 *)(yp + 64 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
}

int main() {
    const int m = 120000;
    float *X, *W, *Y, *Y2;
    hipMalloc(&X, sizeof(float) * (size_t)m * K);
    hipMalloc(&W, sizeof(float) * N * K);
    hipMalloc(&Y, sizeof(float) * (size_t)m * N);
    hipMalloc(&Y2, sizeof(float) * (size_t)m * N);
    std::vector<float> hx((size_t)m * K), hw(N * K);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 1000) / 5000.f - 0.1f;
    hipMemcpy(X, hx.data(), sizeof(float) * hx.size(), hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), sizeof(float) * hw.size(), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto timeit = [&](const char *name, auto fn) {
        for (int i = 0; i < 5; ++i) fn();
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 50; ++i) fn();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("%-28s %7.2f us per launch  (%.2f TB/s of 46 MB)\n", name, 1e3 * ms / 50, 46.08e6 / (1e-3 * ms / 50) / 1e12);
    };
    timeit("oneshot (1875 workgroups)", [&] { hipLaunchKernelGGL(oneshot_kernel, dim3((m + 63) / 64), dim3(256), 0, 0, m, X, W, Y); });
    for (int grid : {256, 512, 768, 1024, 1536, 2048}) {
        char nm[64];
        snprintf(nm, sizeof nm, "stream<1> grid %d", grid);
        timeit(nm, [&] { hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(256), 0, 0, m, X, W, Y2); });
        snprintf(nm, sizeof nm, "stream<2> grid %d", grid);
        timeit(nm, [&] { hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(256), 0, 0, m, X, W, Y2); });
    }
    std::vector<float> y1((size_t)m * N), y2((size_t)m * N);
    hipMemcpy(y1.data(), Y, sizeof(float) * y1.size(), hipMemcpyDeviceToHost);
    hipMemcpy(y2.data(), Y2, sizeof(float) * y2.size(), hipMemcpyDeviceToHost);
    double md = 0;
    for (size_t i = 0; i < y1.size(); ++i) md = std::max(md, (double)fabsf(y1[i] - y2[i]));
    printf("max |oneshot - stream| = %g\n", md);
    return 0;
}
