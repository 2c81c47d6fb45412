// tools/probes/read_pattern_probe.hip -- how fast can a kernel read the (N, G, C) tensor A that the previous kernel wrote?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/read_pattern_probe.hip -o /tmp/rpp && /tmp/rpp
// Patterns (all read every byte once, 12 float4 per lane in flight, 256-thread workgroups, one 64-point block x 3 groups
// per workgroup as peb_fwd_mfma_kernel):
//   strided   lane (point l15, quarter q) walks its quarter of the point's row: 64 distinct 128-B lines per instruction
//   half      lane (l15, q) reads float4 (4 j + q): the 4 lanes of a point read 64 contiguous bytes per instruction
//   coalesced lane t reads float4 (t + 64 i) of the wavefront's 16 rows: 768-byte runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int G = 24, C = 192;
__global__ void write_kernel(float4 *A, long long total4) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256)
        A[e] = make_float4((float)e, 1.f, 2.f, 3.f);
}
template <int MODE>
__global__ __launch_bounds__(256) void read_kernel(int n, const float *__restrict__ A, float *out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l15 = lane & 15, q = lane >> 4;
    const long long pt0 = (long long)blockIdx.x * 64 + wid * 16;
    const int g0 = blockIdx.y * 3;
    float acc = 0.f;
    for (int gl = 0; gl < 3; ++gl) {
        float4 x[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            long long off;
            bool ok;
            if (MODE == 0) { const long long pt = pt0 + l15; ok = pt < n; off = ((pt * G + g0 + gl) * C + q * 48 + 4 * j); }
            else if (MODE == 1) { const long long pt = pt0 + l15; ok = pt < n; off = ((pt * G + g0 + gl) * C + 16 * j + 4 * q); }
            else { const int f = lane + 64 * j; const long long pt = pt0 + f / 48; ok = pt < n; off = ((pt * G + g0 + gl) * C + 4 * (f % 48)); }
            x[j] = ok ? *(const float4 *)(A + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) acc += x[j].x + x[j].y + x[j].z + x[j].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
int main() {
    for (int n : {4501, 18905}) {
        const long long total4 = (long long)n * G * C / 4;
        float *A, *out;
        hipMalloc(&A, total4 * 16);
        hipMalloc(&out, 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const dim3 grid((n + 63) / 64, G / 3);
        for (int mode = 0; mode < 3; ++mode)
            for (int fresh = 1; fresh >= 0; --fresh) {
                float best = 1e9f, sum = 0.f;
                for (int rep = 0; rep < 12; ++rep) {
                    if (fresh) hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(256), 0, 0, (float4 *)A, total4);
                    hipEventRecord(e0, 0);
                    if (mode == 0) hipLaunchKernelGGL(read_kernel<0>, grid, dim3(256), 0, 0, n, A, out);
                    else if (mode == 1) hipLaunchKernelGGL(read_kernel<1>, grid, dim3(256), 0, 0, n, A, out);
                    else hipLaunchKernelGGL(read_kernel<2>, grid, dim3(256), 0, 0, n, A, out);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
                }
                printf("n %6d (%5.1f MB) %-9s %s: best %6.1f us  mean %6.1f us  = %5.2f TB/s\n", n, total4 * 16 / 1e6,
                       mode == 0 ? "strided" : mode == 1 ? "half" : "coalesced", fresh ? "just written" : "re-read    ", best * 1e3, sum / 10 * 1e3,
                       total4 * 16 / (best * 1e-3) / 1e12);
            }
        hipFree(A); hipFree(out);
    }
    return 0;
}
