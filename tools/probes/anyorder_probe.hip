// tools/probes/anyorder_probe.hip -- does hipExtAnyOrderLaunch let two INDEPENDENT kernels of one stream overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/anyorder_probe.hip -o /tmp/anyorder_probe && /tmp/anyorder_probe
// Kernel: WGS workgroups each spin for `us` microseconds of wall clock.  Pairs (A normal, B flagged) against (A, B normal).
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin_kernel(long long ticks, int *sink) {
    const long long t0 = wall_clock64();
    int v = 0;
    while (wall_clock64() - t0 < ticks) v += 1;
    if (v == -1) *sink = v;
}

static double run(int mode, int wgs, double us, int reps, hipStream_t st, int *sink, long long ticks_per_us) {
    const long long ticks = (long long)(us * ticks_per_us);
    auto go = [&](int flags) {
        if (flags) hipExtLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(64), 0, st, nullptr, nullptr, flags, ticks, sink);
        else hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(64), 0, st, ticks, sink);
    };
    hipStreamSynchronize(st);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        go(0);                                   // A: ordered behind everything before it
        go(mode == 1 ? hipExtAnyOrderLaunch : 0);  // B: independent of A
        if (mode == 2) { go(hipExtAnyOrderLaunch); go(hipExtAnyOrderLaunch); }
    }
    hipStreamSynchronize(st);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main() {
    int *sink = nullptr;
    hipMalloc(&sink, 4);
    hipStream_t st;
    hipStreamCreate(&st);
    int rate_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const long long tpu = rate_khz / 1000;  // ticks per microsecond
    printf("wall clock %d kHz\n", rate_khz);
    for (int wgs : {1, 64, 256}) {
        for (double us : {20.0, 100.0}) {
            run(0, wgs, us, 3, st, sink, tpu);
            const double a = run(0, wgs, us, 50, st, sink, tpu), b = run(1, wgs, us, 50, st, sink, tpu), c = run(2, wgs, us, 50, st, sink, tpu);
            printf("wgs %4d spin %5.0f us: A;B ordered %7.1f us   A;B(anyorder) %7.1f us   A;B;C;D(anyorder x3) %7.1f us\n", wgs, us, a, b, c);
        }
    }
    return 0;
}
