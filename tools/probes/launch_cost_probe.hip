// tools/probes/launch_cost_probe.hip -- what does one kernel launch cost the HOST on this box, and what does replaying the
// same sequence as a hipGraph cost?  (the step is host-bound on a slow box; DESIGN 4.1 had looked at the GPU
// side of hipGraph only.)
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_cost_probe tools/probes/launch_cost_probe.hip && /tmp/launch_cost_probe
//
// Prints one JSON object: host microseconds per launch for eager issue (hipLaunchKernelGGL with a 14-argument kernel, the
// shape of the library's launches), for hipMemsetAsync / hipEventRecord, and per node for hipGraphLaunch of a captured chain
// of 650 such launches (the compute-queue launch count of one training step), each with a short (~2 us) and a long (~15 us)
// kernel body, plus the wall time to completion of each form.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <sched.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void body_kernel(long long n, int a, int b, int c, const float *__restrict__ x, const float *__restrict__ w,
                                                   const float *__restrict__ bias, float *__restrict__ y, const int *__restrict__ idx,
                                                   float eps, float mom, float *rec, unsigned *cnt, int spin) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < n) {
        acc = x[i];
        for (int s = 0; s < spin; ++s) acc = __builtin_fmaf(acc, 1.0000001f, eps);
        y[i] = acc + (float)(a + b + c) * mom;
    }
}

__global__ void other_kernel(float *y, int v) { if (blockIdx.x == 0 && threadIdx.x == 0) y[0] = 0.f * v; }

struct Pad { unsigned long long v[126]; };
__global__ void padded_kernel(float *y, Pad pad) { if (threadIdx.x == 0 && pad.v[5] == 77) y[1] = 1.f; }

static double now_us() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e6 * ts.tv_sec + 1e-3 * ts.tv_nsec;
}

struct Res { double host_us_per_launch, wall_us_per_launch; };

int main(int argc, char **argv) {
    const int chain = argc > 1 ? atoi(argv[1]) : 650;
    const int reps = 20;
    const long long n = 256 * 512;
    float *x, *y; unsigned *cnt;
    CK(hipMalloc(&x, sizeof(float) * n)); CK(hipMalloc(&y, sizeof(float) * n)); CK(hipMalloc(&cnt, 256));
    CK(hipMemset(x, 0, sizeof(float) * n));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto launch = [&](int spin) {
        hipLaunchKernelGGL(body_kernel, dim3(512), dim3(256), 0, st, n, 1, 2, 3, (const float *)x, (const float *)x, (const float *)x, y,
                           (const int *)nullptr, 1e-5f, 0.1f, y, cnt, spin);
    };
    auto eager = [&](int spin) {
        for (int i = 0; i < chain; ++i) launch(spin);  // warm
        CK(hipStreamSynchronize(st));
        double host = 0, wall = 0;
        for (int r = 0; r < reps; ++r) {
            double t0 = now_us();
            for (int i = 0; i < chain; ++i) launch(spin);
            double t1 = now_us();
            CK(hipStreamSynchronize(st));
            double t2 = now_us();
            host += t1 - t0; wall += t2 - t0;
        }
        return Res{host / reps / chain, wall / reps / chain};
    };
    auto graph = [&](int spin, double *inst_us) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < chain; ++i) launch(spin);
        CK(hipStreamEndCapture(st, &g));
        double t0 = now_us();
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        *inst_us = now_us() - t0;
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        double host = 0, wall = 0;
        for (int r = 0; r < reps; ++r) {
            double t0 = now_us();
            CK(hipGraphLaunch(ge, st));
            double t1 = now_us();
            CK(hipStreamSynchronize(st));
            double t2 = now_us();
            host += t1 - t0; wall += t2 - t0;
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return Res{host / reps / chain, wall / reps / chain};
    };
    // host facts
    char model[256] = "?";
    if (FILE *f = fopen("/proc/cpuinfo", "r")) {
        char line[512];
        while (fgets(line, sizeof line, f))
            if (!strncmp(line, "model name", 10)) { char *c = strchr(line, ':'); if (c) { strncpy(model, c + 2, 255); model[strcspn(model, "\n")] = 0; } break; }
        fclose(f);
    }
    cpu_set_t set; CPU_ZERO(&set); sched_getaffinity(0, sizeof set, &set);
    double la[3] = {0, 0, 0}; getloadavg(la, 3);
    const char *fdk = getenv("HIP_FORCE_DEV_KERNARG");
    printf("{\"cpu\": \"%s\", \"online\": %ld, \"affinity\": %d, \"loadavg\": [%.2f, %.2f, %.2f], \"HIP_FORCE_DEV_KERNARG\": \"%s\", \"chain\": %d",
           model, sysconf(_SC_NPROCESSORS_ONLN), CPU_COUNT(&set), la[0], la[1], la[2], fdk ? fdk : "", chain);
    for (int spin : {0, 4000}) {
        Res e = eager(spin);
        double inst = 0;
        Res g = graph(spin, &inst);
        printf(", \"spin%d\": {\"eager_host_us\": %.3f, \"eager_wall_us\": %.3f, \"graph_host_us\": %.3f, \"graph_wall_us\": %.3f, \"instantiate_us\": %.1f}",
               spin, e.host_us_per_launch, e.wall_us_per_launch, g.host_us_per_launch, g.wall_us_per_launch, inst);
    }
    // re-capture + hipGraphExecUpdate per step (sizes change every batch), and per-node parameter updates
    {
        hipGraph_t g0; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < chain; ++i) launch(0);
        CK(hipStreamEndCapture(st, &g0));
        CK(hipGraphInstantiate(&ge, g0, nullptr, nullptr, 0));
        double cap = 0, upd = 0, lau = 0, des = 0;
        for (int r = 0; r < reps; ++r) {
            double t0 = now_us();
            hipGraph_t g;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            for (int i = 0; i < chain; ++i)
                hipLaunchKernelGGL(body_kernel, dim3(256 + r + (i & 7)), dim3(256), 0, st, n - r, 1, 2, i, (const float *)x, (const float *)x,
                                   (const float *)x, y, (const int *)nullptr, 1e-5f, 0.1f, y, cnt, 0);
            CK(hipStreamEndCapture(st, &g));
            double t1 = now_us();
            hipGraphExecUpdateResult res; hipGraphNode_t bad;
            CK(hipGraphExecUpdate(ge, g, &bad, &res));
            double t2 = now_us();
            CK(hipGraphLaunch(ge, st));
            double t3 = now_us();
            CK(hipGraphDestroy(g));
            double t4 = now_us();
            CK(hipStreamSynchronize(st));
            cap += t1 - t0; upd += t2 - t1; lau += t3 - t2; des += t4 - t3;
        }
        printf(", \"recapture\": {\"capture_us_per_node\": %.3f, \"exec_update_us_per_node\": %.3f, \"launch_us\": %.1f, \"destroy_us_per_node\": %.3f}",
               cap / reps / chain, upd / reps / chain, lau / reps, des / reps / chain);
        // per-node updates through hipGraphExecKernelNodeSetParams
        size_t nn = 0;
        CK(hipGraphGetNodes(g0, nullptr, &nn));
        hipGraphNode_t *nodes = (hipGraphNode_t *)malloc(sizeof(hipGraphNode_t) * nn);
        CK(hipGraphGetNodes(g0, nodes, &nn));
        double setp = 0;
        int ok = 1;
        for (int r = 0; r < reps; ++r) {
            long long n2 = n - r; int a = 1, b = 2, c = 3, spin = 0; const float *xp = x; float *yp = y; const int *ip = nullptr;
            float eps = 1e-5f, mom = 0.1f; unsigned *cp = cnt;
            void *args[] = {&n2, &a, &b, &c, &xp, &xp, &xp, &yp, &ip, &eps, &mom, &yp, &cp, &spin};
            double t0 = now_us();
            for (size_t i = 0; i < nn; ++i) {
                hipKernelNodeParams kp{};
                kp.func = (void *)body_kernel; kp.gridDim = dim3(300 + r); kp.blockDim = dim3(256); kp.sharedMemBytes = 0;
                kp.kernelParams = args; kp.extra = nullptr;
                if (hipGraphExecKernelNodeSetParams(ge, nodes[i], &kp) != hipSuccess) ok = 0;
            }
            setp += now_us() - t0;
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
        }
        printf(", \"set_params\": {\"us_per_node\": %.3f, \"ok\": %d, \"nodes\": %zu}", setp / reps / chain, ok, nn);
        free(nodes);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g0));
    }
    // (a) are event-record nodes of a captured graph usable for timing?  (b) hipGraphExecUpdate across a changed kernel
    // function, (c) across a changed node count, (d) with memset nodes
    {
        hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
        auto cap = [&](int variant, hipGraph_t *g) {
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            (void)hipMemsetAsync(y, 0, 4096, st);
            launch(0);
            CK(hipEventRecord(ea, st));
            launch(4000);
            CK(hipEventRecord(eb, st));
            if (variant == 1) hipLaunchKernelGGL(other_kernel, dim3(8), dim3(256), 0, st, y, 7);
            else launch(0);
            if (variant == 2) launch(0);
            (void)hipMemsetAsync(y, 0, 8192 + 256 * variant, st);
            CK(hipStreamEndCapture(st, g));
        };
        hipGraph_t g0, g1, g2, g3; hipGraphExec_t ge;
        cap(0, &g0);
        CK(hipGraphInstantiate(&ge, g0, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        float ms = -1.f;
        hipError_t et = hipEventElapsedTime(&ms, ea, eb);
        printf(", \"event_nodes\": {\"elapsed_rc\": %d, \"elapsed_us\": %.2f}", (int)et, 1e3 * ms);
        hipGraphExecUpdateResult res; hipGraphNode_t bad;
        cap(3, &g3);
        hipError_t r3 = hipGraphExecUpdate(ge, g3, &bad, &res);
        printf(", \"update_same_topology\": {\"rc\": %d, \"result\": %d}", (int)r3, (int)res);
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        et = hipEventElapsedTime(&ms, ea, eb);
        printf(", \"event_nodes_after_update\": {\"elapsed_rc\": %d, \"elapsed_us\": %.2f}", (int)et, 1e3 * ms);
        cap(1, &g1);
        hipError_t r1 = hipGraphExecUpdate(ge, g1, &bad, &res);
        printf(", \"update_changed_function\": {\"rc\": %d, \"result\": %d}", (int)r1, (int)res);
        (void)hipGetLastError();
        if (r1 == hipSuccess) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); float v = 0; CK(hipMemcpy(&v, y, 4, hipMemcpyDeviceToHost)); printf(", \"changed_function_ran\": %d", v == 0.f ? 1 : 0); }
        cap(2, &g2);
        hipError_t r2 = hipGraphExecUpdate(ge, g2, &bad, &res);
        printf(", \"update_changed_count\": {\"rc\": %d, \"result\": %d}", (int)r2, (int)res);
        (void)hipGetLastError();
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g0)); CK(hipGraphDestroy(g1)); CK(hipGraphDestroy(g2)); CK(hipGraphDestroy(g3));
    }
    // (e) hipGraphExecUpdate when the kernel at a position changes to one with a LARGER argument block, and to a padded small one
    {
        auto cap2 = [&](int variant, hipGraph_t *g) {
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            launch(0);
            if (variant == 0) hipLaunchKernelGGL(other_kernel, dim3(8), dim3(256), 0, st, y, 7);           // 12-byte arguments
            else if (variant == 1) launch(0);                                                             // ~100-byte arguments
            else { Pad pad{}; hipLaunchKernelGGL(padded_kernel, dim3(1), dim3(64), 0, st, y, pad); }       // 1 KB arguments
            launch(0);
            CK(hipStreamEndCapture(st, g));
        };
        hipGraphExecUpdateResult res; hipGraphNode_t bad;
        for (int from = 0; from < 3; ++from)
            for (int to = 0; to < 3; ++to) {
                if (from == to) continue;
                hipGraph_t ga, gb; hipGraphExec_t ge;
                cap2(from, &ga); cap2(to, &gb);
                CK(hipGraphInstantiate(&ge, ga, nullptr, nullptr, 0));
                hipError_t r = hipGraphExecUpdate(ge, gb, &bad, &res);
                (void)hipGetLastError();
                int ran = 0;
                if (r == hipSuccess) { ran = hipGraphLaunch(ge, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess; }
                printf(", \"update_args_%d_to_%d\": {\"rc\": %d, \"result\": %d, \"ran\": %d}", from, to, (int)r, (int)res, ran);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(ga)); CK(hipGraphDestroy(gb));
            }
    }
    // memset / event record host cost
    {
        double t0 = now_us();
        for (int i = 0; i < 1000; ++i) (void)hipMemsetAsync(y, 0, 4096, st);
        double t1 = now_us();
        CK(hipStreamSynchronize(st));
        hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        double t2 = now_us();
        for (int i = 0; i < 1000; ++i) (void)hipEventRecord(ev, st);
        double t3 = now_us();
        CK(hipStreamSynchronize(st));
        printf(", \"memset_async_host_us\": %.3f, \"event_record_host_us\": %.3f", (t1 - t0) / 1000, (t3 - t2) / 1000);
    }
    printf("}\n");
    return 0;
}
