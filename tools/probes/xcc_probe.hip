// tools/probes/xcc_probe.hip -- which XCD does workgroup b of a launch run on?  (HW_REG_XCC_ID, hwreg 20, bits 3:0)
//   hipcc --offload-arch=gfx950 -O3 -o bin/xcc_probe xcc_probe.hip && bin/xcc_probe [workgroups] [threads]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(int *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 64, threads = argc > 2 ? atoi(argv[2]) : 1024;
    int *d;
    if (hipMalloc(&d, n * sizeof(int)) != hipSuccess) return 1;
    int *scratch;
    if (hipMalloc(&scratch, 64 * sizeof(int)) != hipSuccess) return 1;
    for (int rep = 0; rep < 4; ++rep) {
        // (rep >= 2: a launch of 3 workgroups in front -- does the next launch start where that one stopped?)
        if (rep >= 2) hipLaunchKernelGGL(probe, dim3(3), dim3(threads), 0, 0, scratch);
        hipLaunchKernelGGL(probe, dim3(n), dim3(threads), 0, 0, d);
        std::vector<int> h(n);
        if (hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        printf("launch %d:", rep);
        for (int i = 0; i < n; ++i) printf(" %d", h[i]);
        printf("\n");
    }
    return 0;
}
