// tools/probes/graph_branch_probe.hip -- do parallel branches of a hipGraph run beside each other, and what does an edge between
// branches cost?  (DESIGN.md section 4.5 / profiles/HISTORY.md: the weight-gradient kernels of a Block's backward are off its critical chain.)
//
// A chain of CHAIN dependent small kernels (a few workgroups, a few microseconds each: the deep-level Block backward) and SIDE
// longer low-occupancy kernels (the split-K weight gradients), each of which needs the output of one chain kernel and is needed
// by nobody until the end.  Three graphs, all built by stream capture:
//   serial    every kernel on one stream, the side kernels right behind their producers      (what the library issues today)
//   branch    the side kernels on a second captured stream: one wait per side kernel (on its producer), one join at the end
//   fan       every side kernel on a stream of its own
// Prints the GPU time of one graph launch (events around it, median of REPS) and the same for plain streams + events (eager).
//
//   hipcc --offload-arch=gfx950 -O3 -o bin/graph_branch_probe graph_branch_probe.hip && bin/graph_branch_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// `wgs` workgroups each read a little, wait `ticks` of the 100 MHz clock, write a little
__global__ __launch_bounds__(256) void work_kernel(const float *__restrict__ in, float *__restrict__ out, long long ticks) {
    const long long t0 = wall_clock64();
    float v = in[(blockIdx.x * 256 + threadIdx.x) & 4095];
    while (wall_clock64() - t0 < ticks) v = v * 1.0001f + 0.5f;
    out[(blockIdx.x * 256 + threadIdx.x) & 4095] = v;
}

constexpr int CHAIN = 20, SIDE = 4, REPS = 30;

struct Bufs { float *chain[CHAIN + 1]; float *side[SIDE]; };

static void issue(const Bufs &B, hipStream_t main, hipStream_t *side_streams, int nside_streams, double chain_us, double side_us,
                  int chain_wgs, int side_wgs, std::vector<hipEvent_t> &ev) {
    size_t e = 0;
    int s = 0;
    for (int i = 0; i < CHAIN; ++i) {
        hipLaunchKernelGGL(work_kernel, dim3(chain_wgs), dim3(256), 0, main, (const float *)B.chain[i], B.chain[i + 1], (long long)(chain_us * 100));
        if ((i + 1) % (CHAIN / SIDE) == 0 && s < SIDE) {
            hipStream_t st = main;
            if (nside_streams > 0) {
                st = side_streams[s % nside_streams];
                CK(hipEventRecord(ev[e], main));
                CK(hipStreamWaitEvent(st, ev[e], 0));
                ++e;
            }
            hipLaunchKernelGGL(work_kernel, dim3(side_wgs), dim3(256), 0, st, (const float *)B.chain[i + 1], B.side[s], (long long)(side_us * 100));
            ++s;
        }
    }
    for (int j = 0; j < nside_streams; ++j) {  // join
        CK(hipEventRecord(ev[e], side_streams[j]));
        CK(hipStreamWaitEvent(main, ev[e], 0));
        ++e;
    }
}

static double median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char **argv) {
    const double chain_us = argc > 1 ? atof(argv[1]) : 6.0, side_us = argc > 2 ? atof(argv[2]) : 30.0;
    const int chain_wgs = argc > 3 ? atoi(argv[3]) : 128, side_wgs = argc > 4 ? atoi(argv[4]) : 96;
    Bufs B;
    for (auto &p : B.chain) { CK(hipMalloc(&p, 4096 * 4)); CK(hipMemset(p, 0, 4096 * 4)); }
    for (auto &p : B.side) { CK(hipMalloc(&p, 4096 * 4)); CK(hipMemset(p, 0, 4096 * 4)); }
    hipStream_t main_s, cap, side[SIDE], launch_s;
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&launch_s, hipStreamNonBlocking));
    for (auto &s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(2 * SIDE + 4);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    printf("chain: %d kernels x %.1f us x %d workgroups; side: %d kernels x %.1f us x %d workgroups\n", CHAIN, chain_us, chain_wgs, SIDE,
           side_us, side_wgs);
    printf("sum of durations: chain %.0f us, side %.0f us\n", CHAIN * chain_us, SIDE * side_us);
    const char *names[3] = {"serial", "branch", "fan"};
    const int nstreams[3] = {0, 1, SIDE};
    for (int mode = 0; mode < 3; ++mode) {
        // eager: plain streams and events
        std::vector<float> te;
        for (int r = 0; r < REPS + 3; ++r) {
            CK(hipEventRecord(t0, main_s));
            issue(B, main_s, side, nstreams[mode], chain_us, side_us, chain_wgs, side_wgs, ev);
            CK(hipEventRecord(t1, main_s));
            CK(hipStreamSynchronize(main_s));
            float ms;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (r >= 3) te.push_back(ms * 1e3f);
        }
        // graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
        issue(B, cap, side, nstreams[mode], chain_us, side_us, chain_wgs, side_wgs, ev);
        CK(hipStreamEndCapture(cap, &g));
        size_t nodes = 0;
        CK(hipGraphGetNodes(g, nullptr, &nodes));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        std::vector<float> tg;
        for (int r = 0; r < REPS + 3; ++r) {
            CK(hipEventRecord(t0, launch_s));
            CK(hipGraphLaunch(ge, launch_s));
            CK(hipEventRecord(t1, launch_s));
            CK(hipStreamSynchronize(launch_s));
            float ms;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (r >= 3) tg.push_back(ms * 1e3f);
        }
        printf("%-7s eager %8.1f us   graph %8.1f us   (%zu nodes)\n", names[mode], median(te), median(tg), nodes);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
