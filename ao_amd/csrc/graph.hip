// ao_amd/csrc/graph.hip -- issue a launcher's whole kernel sequence as ONE hipGraph launch (host only).
//
// Why: one training step enqueues ~650 kernels on the compute stream.  Through hipLaunchKernelGGL each costs the launching
// thread 3.5-4.8 us on the pool's hosts (tools/probes/launch_cost_probe.hip: kernarg block written to device memory, AQL
// packet, doorbell), i.e. 2.5-3 ms of a 11 ms step before any launcher logic -- and on a busy host the step becomes
// host-bound (BENCH_r03: 13.7 ms for 10.9 ms of kernels).  The same probe: capturing a launch costs 0.65-0.72 us,
// hipGraphExecUpdate 0.9-1.0 us per node, hipGraphLaunch of 650 nodes 11 us in total (the runtime writes the AQL packets
// as one batch), and the GPU runs the batch with 1.7 us per empty kernel instead of 3.6.
//
// How: the model launchers (model.hip) open a PtvGraphScope on the caller's stream.  The scope puts the stream into
// capture mode (a stream of the library's own per caller stream: torch's default stream is the null stream, which cannot
// capture), the launcher body runs unchanged on scope.stream() (its hipLaunchKernelGGL calls become kernel nodes), and
// `finish` ends the capture, brings the executable graph of that (device, stream, slot) up to date and launches it on the
// caller's stream:
//   * same node count as the previous call -> hipGraphExecUpdate (arguments, grids AND kernel functions may differ: level
//     sizes change with every batch, and with them the grids and the tile shapes the launchers pick);
//   * different node count -> instantiate a new executable graph (1.4-1.9 us per node, once per topology change).
// Nothing is replayed blindly: every call re-derives every argument from the caller's struct, so a graph launch enqueues
// exactly what the eager path would have.  The scope declines (the launcher then runs eagerly on the caller's stream) when
// that stream is already capturing (a caller's own torch.cuda.graph), when the launcher asks it to (cross-stream events
// in the body), when the HIP-event kernel timer brackets every
// kernel (event-record nodes cannot be read with hipEventElapsedTime: rc 400 in the probe), or with AO_AMD_GRAPH=0.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace {

struct SlotKey {
    int device;
    hipStream_t stream;
    int which;
    bool operator==(const SlotKey &o) const { return device == o.device && stream == o.stream && which == o.which; }
};
struct SlotKeyHash {
    size_t operator()(const SlotKey &k) const {
        return std::hash<void *>()((void *)k.stream) * 131u + (size_t)k.device * 7u + (size_t)k.which;
    }
};
// An executable graph keeps its kernel arguments in device memory of its own, and the host may run a full step ahead of the
// GPU: updating the one graph that the previous step's launch is still executing from would rewrite arguments under
// running kernels.  So every slot holds a small ring of executable graphs, each fenced by an event recorded behind its
// launch; a ring entry is only updated (or destroyed) once that event has completed -- which also bounds how far the
// host can run ahead (RING steps).  Eight: the GPU boxes of this pool run under a cgroup CPU quota (cpu.max 16 CPUs per
// 100 ms on a 256-thread host), and a throttled period stalls the issuing thread for up to ~90 ms; with eager issue the
// host is only ~3 ms ahead of the GPU and every such stall is a GPU bubble, eight queued steps (~90 ms) ride it out.
constexpr int RING = 8;
struct Exec {
    hipGraphExec_t exec = nullptr;
    size_t nodes = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
    unsigned long long seq = 0;  // launch order, to find the oldest
};
struct Slot {
    Exec ring[RING];
    unsigned long long launches = 0;
};
// the entry the next launch of a slot uses: the first whose previous launch has completed (an entry is created -- one
// instantiation -- only when every existing one is still in flight, i.e. only as far as the host actually runs ahead),
// else the oldest, which finish() then waits for
int pick_entry(Slot &slot) {
    int oldest = 0;
    for (int i = 0; i < RING; ++i) {
        Exec &e = slot.ring[i];
        if (e.in_flight && e.done && hipEventQuery(e.done) == hipSuccess) e.in_flight = false;
        if (!e.in_flight) return i;
        if (e.seq < slot.ring[oldest].seq) oldest = i;
    }
    (void)hipGetLastError();  // (hipErrorNotReady of the queries)
    return oldest;
}
struct StreamKey {
    int device;
    hipStream_t stream;
    bool operator==(const StreamKey &o) const { return device == o.device && stream == o.stream; }
};
struct StreamKeyHash {
    size_t operator()(const StreamKey &k) const { return std::hash<void *>()((void *)k.stream) * 131u + (size_t)k.device; }
};
std::unordered_map<StreamKey, hipStream_t, StreamKeyHash> g_capture_streams;  // caller stream -> the stream its bodies are captured on

std::mutex g_mu;
std::unordered_map<SlotKey, Slot, SlotKeyHash> g_slots;
// the fence event of the most recent graph launched on a caller stream (owned by its ring entry): completed = that stream has
// run dry of this library's graphs, i.e. the host is NOT ahead of the GPU (a training loop that synchronises every step)
std::unordered_map<StreamKey, hipEvent_t, StreamKeyHash> g_last_done;
std::atomic<int> g_mode{-1};  // -1: read AO_AMD_GRAPH on first use; 0 off; 1 on
thread_local int g_inside = 0;  // this thread's launches are being captured (the kernel timer stamps instead of recording events)

// counters for bench.py's `host` object
struct Stats {
    std::atomic<long long> scopes{0}, updated{0}, instantiated{0}, declined{0}, nodes{0};
    std::atomic<long long> capture_ns{0}, update_ns{0}, launch_ns{0}, wait_ns{0};
} g_stats;

long long now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int mode() {
    int m = g_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char *e = getenv("AO_AMD_GRAPH");
        m = (e && e[0] == '0') ? 0 : 1;
        g_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}

}  // namespace

int ptv2_graph_capturing(void) { return g_inside; }
void ptv2_profile_scope(int which, int end, int ring_entry, int ring_size);  // abi.hip

namespace {
__global__ __launch_bounds__(256) void zero_kernel(float4 *__restrict__ p, long long n4, float *__restrict__ tail, int ntail) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}
}  // namespace

int ptv2_zero_async(void *p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return PTV2_OK;
    if (!p || (bytes & 3) || ((uintptr_t)p & 15)) return PTV2_ERR_ARG;
    const long long n4 = (long long)(bytes >> 4);
    const int ntail = (int)((bytes & 15) >> 2);
    const int grid = (int)std::min<long long>(std::max<long long>((n4 + 255) / 256, 1), 2048);
    hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, st, (float4 *)p, n4, (float *)p + 4 * n4, ntail);
    return PTV2_OK;
}

// on: 0 / 1; negative: leave unchanged.  Returns the previous setting.
extern "C" int ptv2_graph_mode(int on) {
    const int prev = mode();
    if (on >= 0) g_mode.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}

// out[0..8] = scopes opened, graphs updated in place, graphs instantiated, scopes declined (ran eagerly), nodes launched,
// host microseconds spent capturing (the launcher bodies included), updating / instantiating, launching, waiting for a ring entry
// whose previous launch the GPU has not finished.  reset != 0 zeroes them.
extern "C" int ptv2_graph_stats(double *out, int reset) {
    if (out) {
        out[0] = (double)g_stats.scopes.load(); out[1] = (double)g_stats.updated.load();
        out[2] = (double)g_stats.instantiated.load(); out[3] = (double)g_stats.declined.load();
        out[4] = (double)g_stats.nodes.load(); out[5] = 1e-3 * (double)g_stats.capture_ns.load();
        out[6] = 1e-3 * (double)g_stats.update_ns.load(); out[7] = 1e-3 * (double)g_stats.launch_ns.load();
        out[8] = 1e-3 * (double)g_stats.wait_ns.load();
    }
    if (reset) {
        g_stats.scopes = 0; g_stats.updated = 0; g_stats.instantiated = 0; g_stats.declined = 0; g_stats.nodes = 0;
        g_stats.capture_ns = 0; g_stats.update_ns = 0; g_stats.launch_ns = 0; g_stats.wait_ns = 0;
    }
    return PTV2_OK;
}

// drop every executable graph (tests; a process that is about to destroy its streams)
extern "C" int ptv2_graph_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_slots)
        for (Exec &e : kv.second.ring) {
            if (e.in_flight && e.done) (void)hipEventSynchronize(e.done);
            if (e.exec) (void)hipGraphExecDestroy(e.exec);
            if (e.done) (void)hipEventDestroy(e.done);
        }
    g_slots.clear();
    g_last_done.clear();
    for (auto &kv : g_capture_streams) (void)hipStreamDestroy(kv.second);
    g_capture_streams.clear();
    return PTV2_OK;
}

PtvGraphScope::PtvGraphScope(void *stream, int which_, bool allow)
    : st(stream), cap(stream), which(which_), ring_entry(0), active(false), t0(0) {
    hipStream_t s = (hipStream_t)stream;
    if (!allow || !mode() || g_inside) { g_stats.declined++; return; }
    // HIP-event brackets around every kernel (the survey of bench.py's roofline leg) cannot live in a graph
    if (ptv2_profile_is_on() && !ptv2_profile_stamps()) { g_stats.declined++; return; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s != nullptr && (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)) {
        g_stats.declined++; (void)hipGetLastError(); return;
    }
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { g_stats.declined++; return; }
    if (which == GRAPH_MODEL_FWD_PREFIX || which == GRAPH_MODEL_FWD) {
        // The first launches of a step behind an idle GPU: capturing, updating and launching the forward's (or its prefix's)
        // graph keeps the GPU waiting for 0.3-0.8 ms of host work; issued eagerly its first kernel starts microseconds after
        // the call (a loop that reads the loss back every iteration -- pointcept's InformationWriter -- pays that at every
        // step, any loop at its first step behind a synchronisation: the first timed step of the bench was 11.6 ms against
        // 9.85).  Same kernels, same arguments, same order either way.  AO_AMD_GRAPH_IDLE_EAGER=0: always the graph.
        const char *e = getenv("AO_AMD_GRAPH_IDLE_EAGER");
        if (!(e && e[0] == '0')) {
            hipEvent_t last = nullptr;
            {
                std::lock_guard<std::mutex> lk(g_mu);
                auto it = g_last_done.find(StreamKey{device, s});
                if (it != g_last_done.end()) last = it->second;
            }
            if (last && hipEventQuery(last) == hipSuccess) return;  // (not counted as declined: a choice, not a refusal)
            (void)hipGetLastError();
        }
    }
    hipStream_t c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_capture_streams.find(StreamKey{device, s});
        if (it != g_capture_streams.end()) {
            c = it->second;
        } else {
            if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) { g_stats.declined++; (void)hipGetLastError(); return; }
            // first use of a stream allocates and zeroes its arrival counters (abi.hip): outside the capture (the memset would
            // be replayed), and complete before the first graph that uses them runs on the caller's stream
            if (!ptv2_stream_counters(c) || hipStreamSynchronize(c) != hipSuccess) {
                (void)hipStreamDestroy(c); g_stats.declined++; return;
            }
            g_capture_streams[StreamKey{device, s}] = c;
        }
    }
    int entry = 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        entry = pick_entry(g_slots[SlotKey{device, s, which}]);  // (one issuing thread per stream: finish() uses this entry)
    }
    t0 = now_ns();
    if (hipStreamBeginCapture(c, hipStreamCaptureModeRelaxed) != hipSuccess) { g_stats.declined++; (void)hipGetLastError(); return; }
    cap = (void *)c;
    active = true;
    g_inside = 1;
    g_stats.scopes++;
    ring_entry = entry;
    ptv2_profile_scope(which, 0, entry, RING);
}

int PtvGraphScope::finish(int rc) {
    if (!active) return rc;
    active = false;
    ptv2_profile_scope(which, 1, 0, RING);
    g_inside = 0;
    hipStream_t s = (hipStream_t)st;
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture((hipStream_t)cap, &g);
    const long long t1 = now_ns();
    g_stats.capture_ns += t1 - t0;
    if (ec != hipSuccess || !g) { (void)hipGetLastError(); if (g) (void)hipGraphDestroy(g); return rc != PTV2_OK ? rc : PTV2_ERR_LAUNCH; }
    if (rc != PTV2_OK) { (void)hipGraphDestroy(g); return rc; }  // the launcher refused half-way: nothing was enqueued
    int device = 0;
    (void)hipGetDevice(&device);
    size_t nodes = 0;
    (void)hipGraphGetNodes(g, nullptr, &nodes);
    int out = PTV2_OK;
    // g_mu protects the slot TABLE only: the ring entry is looked up under it, the wait for its previous launch, the update and
    // the launch run outside (they can take a whole GPU step; a second issuing thread -- another model, an evaluation pass on its
    // own stream -- must not queue behind them).  One issuing thread per (device, stream, direction) is the contract of the
    // scope (the entry was picked by the constructor of THIS scope); entries of a std::unordered_map stay where they are.
    Exec *slot_p = nullptr;
    Slot *ring_p = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        Slot &ring = g_slots[SlotKey{device, s, which}];
        ring_p = &ring;
        slot_p = &ring.ring[ring_entry];
        slot_p->seq = ++ring.launches;
    }
    {
        Exec &slot = *slot_p;
        // (blocking: the wait for a ring entry -- the host eight steps ahead of the GPU -- sleeps instead of spinning; eight
        // ranks of a node share one CPU quota, and a spinning launching thread was a whole CPU per rank)
        if (!slot.done && hipEventCreateWithFlags(&slot.done, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) slot.done = nullptr;
        if (slot.in_flight && slot.done) {
            const long long w0 = now_ns();
            (void)hipEventSynchronize(slot.done);  // (the host is RING steps ahead of the GPU)
            g_stats.wait_ns += now_ns() - w0;
            slot.in_flight = false;
        }
        {   // AO_AMD_GRAPH_LEAD = launches in flight per slot (default RING = 8: the host may run eight steps ahead).  Measured:
            // behind a synchronisation the host issues eight steps' graphs in a burst, and the GPU runs the two or three steps
            // it executes meanwhile 0.3-0.4 ms slower (10.2 against 9.8 ms); with 3 in flight those steps run at 9.8 and a
            // 20-step bench line reads 0.01-0.03 ms lower.  A training run has no such burst, and what the depth buys -- 80 ms
            // of cover against a host thread that is descheduled (a cgroup quota period is 100 ms) -- matters more: the
            // default stays at the ring's depth.
            static const int lead = [] { const char *e = getenv("AO_AMD_GRAPH_LEAD"); const int v = e ? atoi(e) : RING; return v < 1 ? 1 : (v > RING ? RING : v); }();
            if (lead < RING) {
                for (;;) {
                    int inflight = 0; Exec *oldest = nullptr;
                    for (int e2 = 0; e2 < RING; ++e2) {
                        Exec &o = ring_p->ring[e2];
                        if (&o == &slot || !o.in_flight || !o.done) continue;
                        if (hipEventQuery(o.done) == hipSuccess) { o.in_flight = false; continue; }
                        ++inflight;
                        if (!oldest || o.seq < oldest->seq) oldest = &o;
                    }
                    (void)hipGetLastError();
                    if (inflight < lead || !oldest) break;
                    const long long w0 = now_ns();
                    (void)hipEventSynchronize(oldest->done);
                    g_stats.wait_ns += now_ns() - w0;
                    oldest->in_flight = false;
                }
            }
        }
        const long long t1b = now_ns();
        bool ready = false;
        static const bool debug = [] { const char *e = getenv("AO_AMD_GRAPH_DEBUG"); return e && e[0] == '1'; }();
        if (slot.exec && slot.nodes == nodes) {
            hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
            hipGraphNode_t bad = nullptr;
            const hipError_t eu = hipGraphExecUpdate(slot.exec, g, &bad, &res);
            if (eu == hipSuccess && res == hipGraphExecUpdateSuccess) {
                ready = true;
                g_stats.updated++;
            } else {
                (void)hipGetLastError();
                if (debug) fprintf(stderr, "ptv2 graph: slot %d update failed rc %d result %d (%zu nodes)\n", which, (int)eu, (int)res, nodes);
            }
        } else if (debug && slot.exec) {
            fprintf(stderr, "ptv2 graph: slot %d node count %zu -> %zu\n", which, slot.nodes, nodes);
        }
        if (!ready) {
            if (slot.exec) { (void)hipGraphExecDestroy(slot.exec); slot.exec = nullptr; }
            if (hipGraphInstantiate(&slot.exec, g, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGetLastError();
                slot.exec = nullptr;
                out = PTV2_ERR_LAUNCH;
            } else {
                slot.nodes = nodes;
                g_stats.instantiated++;
                // The ring entries that have never been used are instantiated from the same capture right away: left to their
                // first use, a loop's steps 2 .. RING each paid two instantiations (forward, backward) on the launching thread
                // while the GPU caught up with it -- the first five timed steps of the bench ran 10.0-10.3 ms against 9.85
                // (five warm-up steps fill five of eight entries).  Their first use then is an in-place update like any other.
                // (Not while the kernel timer's stamps are on: the bracket positions differ from entry to entry.)
                if (!ptv2_profile_is_on()) {
                    for (int e = 0; e < RING; ++e) {
                        Exec &o = ring_p->ring[e];
                        if (&o == &slot || o.exec || o.in_flight) continue;
                        if (hipGraphInstantiate(&o.exec, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); o.exec = nullptr; break; }
                        o.nodes = nodes;  // (not counted: `instantiated` + `updated` = scopes launched)
                    }
                }
            }
        }
        const long long t2 = now_ns();
        g_stats.update_ns += t2 - t1b;
        if (out == PTV2_OK) {
            if (hipGraphLaunch(slot.exec, s) != hipSuccess) { (void)hipGetLastError(); out = PTV2_ERR_LAUNCH; }
            else if (slot.done && hipEventRecord(slot.done, s) == hipSuccess) {
                slot.in_flight = true;
                std::lock_guard<std::mutex> lk(g_mu);
                g_last_done[StreamKey{device, s}] = slot.done;
            }
            else if (hipStreamSynchronize(s) != hipSuccess) out = PTV2_ERR_LAUNCH;  // no fence: wait here instead
            g_stats.launch_ns += now_ns() - t2;
            g_stats.nodes += (long long)nodes;
        }
    }
    (void)hipGraphDestroy(g);
    return out;
}

PtvGraphScope::~PtvGraphScope() {
    if (!active) return;  // finished, or never started
    // an early return of the launcher body: close the capture and discard it
    ptv2_profile_scope(which, 1, 0, RING);  // (the kernel timer's per-scope state must not outlive the aborted scope)
    g_inside = 0;
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture((hipStream_t)cap, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
}
