// tools/probes/mfma_rate_probe.hip -- calibration: how long does v_mfma_f32_16x16x4_f32 take per instruction on this box, as a
// function of the number of independent accumulator chains per wavefront and of wavefronts per SIMD?
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate_probe.hip -o tools/probes/bin/mfma_rate_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void probe(float *out, int iters) {
    v4f acc[CH];
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
void run(float *d, int wg_per_cu) {
    const int iters = 4096 / CH * 4, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<CH><<<grid, 256>>>(d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<CH><<<grid, 256>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * CH * wg_per_cu;  // one wavefront of each workgroup per SIMD
    printf("chains %d, %d wavefront(s) per SIMD: %.1f us, %.1f ns per MFMA per SIMD (32 cycles at 2.4 GHz = 13.3 ns)\n", CH, wg_per_cu,
           ms * 1e3, ms * 1e6 / mfma_per_simd);
}
int main() {
    float *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w = 1; w <= 2; ++w) { run<1>(d, w); run<2>(d, w); run<4>(d, w); run<8>(d, w); }
    return 0;
}
