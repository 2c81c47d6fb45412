// ao_amd/csrc/abi.hip -- library identification and the optional per-kernel timer (host only).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "gva_common.h"

extern "C" int ptv2_abi_version(void) { return 11; }  // == EXPECTED_ABI in ao_amd/_lib.py

// sizeof() of the structs that ctypes mirrors field by field (block.py, native_model.py): compared at load time, so a
// layout drift between the header and a python mirror is an import error, not a misread pointer
extern "C" long long ptv2_struct_bytes(int which) {
    switch (which) {
        case 0: return (long long)sizeof(ptv2_block);
        case 1: return (long long)sizeof(ptv2_block_grads);
        case 2: return (long long)sizeof(ptv2_model);
        case 3: return (long long)sizeof(ptv2_gva_block);
        case 4: return (long long)sizeof(ptv2_scene_geo);
        default: return -1;
    }
}

namespace { thread_local int g_matmul_bf16 = 0; }
int ptv2_matmul_bf16(void) { return g_matmul_bf16; }
void ptv2_set_matmul_bf16(int on) { g_matmul_bf16 = on ? 1 : 0; }
// for callers of the stand-alone launchers (rows_gemm_*, linear_wgrad_*): returns the previous setting of this thread
extern "C" int ptv2_matmul_precision(int bf16) {
    const int prev = g_matmul_bf16;
    if (bf16 >= 0) g_matmul_bf16 = bf16 ? 1 : 0;
    return prev;
}

// ---- attention dropout of the gva_block call in progress on this thread (gva_common.h)
namespace gva {
namespace { thread_local PtvDrop g_drop{1.f, 0u, 0u}; }
PtvDrop ptv2_attn_drop_current() { return g_drop; }
void ptv2_attn_drop_set(float p, unsigned seed) {
    if (!(p > 0.f)) { g_drop = PtvDrop{1.f, 0u, 0u}; return; }
    const double pp = p < 1.f ? (double)p : 1.0;
    const double t = pp * 4294967296.0;
    g_drop.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    if (g_drop.thresh == 0u) g_drop.thresh = 1u;
    g_drop.scale = pp < 1.0 ? (float)(1.0 / (1.0 - pp)) : 0.f;
    g_drop.seed = seed;
}
PtvAttnDropScope::~PtvAttnDropScope() { g_drop = prev; }
}  // namespace gva

// ---- riders (gva_common.h): per-thread queue of finalizes that a later, independent launch carries
namespace gva {
namespace {
thread_local int g_defer_depth = 0;
thread_local PtvRiders g_pending{};
__global__ __launch_bounds__(256) void rider_kernel(PtvRiders Rs) { rider_run(Rs, blockIdx.x); }
void launch_alone(const PtvRider &r, hipStream_t st) {
    PtvRiders one{};
    one.count = 1;
    one.r[0] = r;
    hipLaunchKernelGGL(rider_kernel, dim3(r.blocks), dim3(256), 0, st, one);
}
}  // namespace
bool ptv2_rider_defer_active() {
    static const bool off = [] { const char *e = getenv("AO_AMD_RIDERS"); return e && e[0] == '0'; }();  // A/B switch (tests)
    return g_defer_depth > 0 && !off;
}
void ptv2_rider_defer_depth(int delta) { g_defer_depth += delta; }
void ptv2_rider_defer(const PtvRider &r, hipStream_t st) {
    if (g_pending.count == RIDER_QUEUE) {  // full: the oldest goes out on its own
        launch_alone(g_pending.r[0], st);
        for (int i = 1; i < RIDER_QUEUE; ++i) g_pending.r[i - 1] = g_pending.r[i];
        g_pending.count = RIDER_QUEUE - 1;
    }
    g_pending.r[g_pending.count++] = r;
}
PtvRiders ptv2_rider_take() {
    const PtvRiders out = g_pending;
    g_pending.count = 0;
    return out;
}
void ptv2_rider_flush(hipStream_t st) {
    for (int i = 0; i < g_pending.count; ++i) launch_alone(g_pending.r[i], st);
    g_pending.count = 0;
}
int ptv2_rider_drop() {
    const int had = g_pending.count;
    g_pending.count = 0;
    return had;
}
}  // namespace gva

#ifndef PTV2_SRC_HASH
#define PTV2_SRC_HASH "unknown"
#endif

extern "C" const char *ptv2_build_info(void) {
    return "libptv2_hip gfx950 (MI355X) src " PTV2_SRC_HASH " hipcc " __VERSION__ " built " __DATE__;
}

// ------------------------------------------------------------ logit basket (host) --
// dst[ids[r], :] = src[r, :] for r in [0, rows): the trainer statement `basket[k][ori_idx] = seg`
// (pointcept/engines/train_sam_real.py:234) on host memory, called by ao_amd/ptv2/basket.py's worker thread through
// ctypes (which drops the interpreter lock for the call, so the training thread keeps launching kernels).  Rows are
// written in ascending r (a repeated id keeps its LAST row, as numpy's assignment does).  Returns PTV2_ERR_ARG, writing
// nothing, if an id is outside [0, dst_rows).
extern "C" int basket_scatter_rows_host(float *dst, long long dst_rows, const long long *ids, const float *src, long long rows,
                                        int c) {
    if (!dst || !ids || !src || rows < 0 || c < 1 || dst_rows < 0) return PTV2_ERR_ARG;
    for (long long r = 0; r < rows; ++r)
        if (ids[r] < 0 || ids[r] >= dst_rows) return PTV2_ERR_ARG;
    for (long long r = 0; r < rows; ++r) memcpy(dst + ids[r] * c, src + r * c, sizeof(float) * (size_t)c);
    return PTV2_OK;
}

// ------------------------------------------------------- per-stream arrival counters --
// Kernels that finish their own per-block partial sums ("last block done", gva_common.h) need a zeroed device
// counter; the last block resets it, so one small array per stream lives for the life of the library (kernels
// on one stream are serialised, each launcher uses its own slot).  Keyed by (device, stream): the default stream
// has the same handle (0) on every device, and the array must live on the device the kernel runs on.
namespace {
std::mutex g_cnt_mu;
struct CounterKey {
    int device;
    hipStream_t stream;
    bool operator==(const CounterKey &o) const { return device == o.device && stream == o.stream; }
};
struct CounterKeyHash {
    size_t operator()(const CounterKey &k) const { return std::hash<void *>()((void *)k.stream) * 31u + (size_t)k.device; }
};
std::unordered_map<CounterKey, unsigned *, CounterKeyHash> g_counters;
}  // namespace

unsigned *ptv2_stream_counters(hipStream_t st) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_cnt_mu);
    const CounterKey key{device, st};
    auto it = g_counters.find(key);
    if (it != g_counters.end()) return it->second;
    unsigned *p = nullptr;
    if (hipMalloc((void **)&p, sizeof(unsigned) * PTV2_NUM_COUNTERS) != hipSuccess) return nullptr;
    // zeroed ON the stream that will use them: a null-stream hipMemset is not ordered against a non-blocking stream (every
    // torch side stream is one), and the first kernel there could have met the allocation's previous contents
    if (hipMemsetAsync(p, 0, sizeof(unsigned) * PTV2_NUM_COUNTERS, st) != hipSuccess) return nullptr;
    g_counters[key] = p;
    return p;
}

// ---------------------------------------------------------------- kernel timer --
// A debugging / measurement facility, off by default: no events are created and launchers take no extra
// branch beyond one flag read.  Enabled by bench.py for its roofline object (HIP events on the launch
// stream around ONE named kernel per launcher, so the averages are comparable with rocprofv3 --stats).
namespace {
// A name that ends in " [family]" brackets several kernel symbols (instantiations of one template chosen by shape, or the
// kernels of one BatchNorm pass): bench.py ranks those as families and takes its dominant KERNEL among the others, each of
// which is one symbol of rocprofv3's table.
const char *kNames[KID_COUNT] = {
    "knn_row_query_kernel [family]", "logits_fwd_kernel [family]", "softmax_rows_kernel [family]", "aggregate_tile_kernel [family]",
    "peb_fwd_kernel [family]", "peb_bwd_kernel<8>", "aggregate_bwd_tile_kernel [family]", "aggregate_bwd_rows_kernel [family]",
    "aggregate_bwd_gv_kernel<8>", "logits_bwd_rows_kernel [family]", "logits_bwd_gather_kernel", "logits_bwd_params_kernel [family]",
    "linear_wgrad_kernel [family]", "bn_stats_kernel", "bn_apply_kernel [family]", "bn_bwd_reduce_kernel [family]",
    "bn_bwd_apply_kernel [family]", "skinny_fwd_kernel", "skinny_bwd_kernel", "rows_gemm_kernel<48, false, 32> [family]",
    "rows_gemm_kernel<48, false, 64> [family]", "rows_gemm_kernel<48, true, 32> [family]", "rows_gemm_kernel<48, true, 64> [family]",
    "rows_gemm_kernel<64, false, 32> [family]", "rows_gemm_kernel<64, false, 64> [family]", "rows_gemm_kernel<64, true, 32> [family]",
    "rows_gemm_kernel<64, true, 64> [family]",
    "attention_bwd_point_kernel<6, 48, 1, true>",
    "attention_bwd_point_kernel<12, 96, 1, false>", "attention_bwd_point_kernel<24, 192, 2, false>",
    "attention_bwd_point_kernel<48, 384, 4, false>", "attention_bwd_point_kernel<64, 512, 4, false>",
    "linear_wgrad_lds_kernel [family]",  // (both instances and their batched form)
    "grouped_wgrad_kernel_jobs", "attention_fwd_point6_kernel",
    "logits_bwd_fused6_kernel<48>", "logits_bwd_fused_kernel<12, 96, 1>", "logits_bwd_fused_kernel<24, 192, 4>",
    "logits_bwd_fused_kernel<48, 384, 4>", "logits_bwd_fused_kernel<64, 512, 4>",
    "bn_bwd_apply_residual_kernel", "bn_bwd_finapply_kernel [family]",
    "attention_fwd_tile_kernel<12, 96, 12>", "attention_fwd_tile_kernel<24, 192, 12>", "attention_fwd_tile_kernel<48, 384, 12>",
    "attention_fwd_tile_kernel<64, 512, 16>", "wp2_wgrad_tile_kernel_jobs",
    "attention_bwd_tile_kernel<12, 96, 8>", "attention_bwd_tile_kernel<24, 192, 8>", "attention_bwd_tile_kernel<48, 384, 8>",
    "attention_bwd_tile_kernel<64, 512, 8>"};
struct Rec { hipEvent_t a, b; double bytes; };
std::mutex g_mu;
int g_on = 0;
int g_only = -1;                 // >= 0: time this kernel id only (ptv2_profile_select)
int g_stride = 1;                // bracket every g_stride-th launch of a timed kernel (ptv2_profile_stride)
unsigned g_seen[KID_COUNT];      // launches seen per kernel id since enable
std::vector<hipEvent_t> g_free;  // recycled events: creating one per launch costs more than recording it
hipEvent_t take_event() {
    if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
std::vector<Rec> g_recs[KID_COUNT];
std::vector<hipEvent_t> g_pending[KID_COUNT];
// ---- brackets inside a captured launcher body (graph.hip): device time stamps instead of HIP events.
// An event-record node of a graph cannot be read with hipEventElapsedTime (rc 400, tools/probes/launch_cost_probe.hip), so a
// bracket there is [stamp kernel, timed kernel, stamp kernel]: one lane stores wall_clock64() (the constant-rate counter,
// hipDeviceAttributeWallClockRate kHz: 100 MHz on gfx950).  A captured sequence is brought into the executable graph of
// its ring entry by hipGraphExecUpdate, which needs the same node count and -- measured -- refuses a node whose kernel
// moves to another code object ("not supported", result 6): the brackets of one ring entry therefore sit at the SAME launch
// positions every time, launch ordinal o of the timed kernel inside the scope being bracketed iff ((o + phase) % P) %
// stride == 0 with P = the kernel's launch count in the previous scope of that slot and phase = ring entry * stride / ring
// size (so the ring's entries together sample ring-size times as many positions as one).
constexpr int STAMP_CAP = 1 << 15;  // brackets per enable
unsigned long long *g_stamp_buf = nullptr;  // device, 2 * STAMP_CAP
double g_stamp_us_per_tick = 0.01;
int g_stamp_n = 0;
struct StampRec { int slot; double bytes; };
std::vector<StampRec> g_stamp_recs[KID_COUNT];
std::vector<int> g_stamp_pending[KID_COUNT];
unsigned g_scope_period[GRAPH_SLOTS][KID_COUNT];  // launches of each kernel in the previous scope of a slot
unsigned g_scope_phase[GRAPH_SLOTS];
thread_local unsigned t_scope_seen[KID_COUNT];
thread_local int t_scope_which = -1;
__global__ void stamp_kernel(unsigned long long *p) {
    if (threadIdx.x == 0) *p = wall_clock64();
}
}  // namespace

extern "C" int ptv2_profile_is_on(void) { return g_on; }
int ptv2_profile_stamps(void) { return g_on && g_only >= 0 && g_stamp_buf != nullptr; }
void ptv2_profile_scope(int which, int end, int ring_entry, int ring_size) {  // graph.hip: a captured body begins / ends on this thread
    if (which < 0 || which >= GRAPH_SLOTS) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!end) {
        for (int k = 0; k < KID_COUNT; ++k) t_scope_seen[k] = 0;
        t_scope_which = which;
        const int stride = g_stride > 1 ? g_stride : 1;
        g_scope_phase[which] = (unsigned)(ring_entry * std::max(1, stride / std::max(1, ring_size)));
    } else {
        for (int k = 0; k < KID_COUNT; ++k) g_scope_period[which][k] = t_scope_seen[k];
        t_scope_which = -1;
    }
}
int ptv2_profile_wants(int kid) {
    if (!g_on || (g_only >= 0 && g_only != kid)) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    if (ptv2_graph_capturing() && t_scope_which >= 0) {
        const unsigned o = t_scope_seen[kid]++, P = g_scope_period[t_scope_which][kid], ph = g_scope_phase[t_scope_which];
        const unsigned stride = g_stride > 1 ? (unsigned)g_stride : 1u;
        return (P > 0 ? ((o + ph) % P) % stride : o % stride) == 0;
    }
    if (g_stride <= 1) return 1;
    return (g_seen[kid]++ % (unsigned)g_stride) == 0;
}
// bracket only every n-th launch of the timed kernels (n >= 1): a uniform sample of the launches in a timed region
extern "C" int ptv2_profile_stride(int n) {
    if (n < 1) return PTV2_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    g_stride = n;
    return PTV2_OK;
}

// kid >= 0: only that kernel is bracketed with events from now on (keeps the timer out of the way of a
// whole-step measurement); kid < 0: all kernels
extern "C" int ptv2_profile_select(int kid) {
    if (kid >= KID_COUNT) return PTV2_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    g_only = kid < 0 ? -1 : kid;
    return PTV2_OK;
}

void ptv2_profile_begin(int kid, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (ptv2_graph_capturing()) {
        if (!g_stamp_buf || g_stamp_n >= STAMP_CAP) return;
        const int slot = g_stamp_n++;
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, st, g_stamp_buf + 2 * slot);
        g_stamp_pending[kid].push_back(slot);
        return;
    }
    hipEvent_t e = take_event();
    if (!e) return;
    (void)hipEventRecord(e, st);
    g_pending[kid].push_back(e);
}

void ptv2_profile_end(int kid, hipStream_t st, double bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (ptv2_graph_capturing()) {
        if (g_stamp_pending[kid].empty()) return;
        const int slot = g_stamp_pending[kid].back();
        g_stamp_pending[kid].pop_back();
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, st, g_stamp_buf + 2 * slot + 1);
        g_stamp_recs[kid].push_back(StampRec{slot, bytes});
        return;
    }
    hipEvent_t e = take_event();
    if (!e) return;
    (void)hipEventRecord(e, st);
    if (g_pending[kid].empty()) { g_free.push_back(e); return; }
    hipEvent_t a = g_pending[kid].back();
    g_pending[kid].pop_back();
    g_recs[kid].push_back(Rec{a, e, bytes});
}

// on != 0: clear the table and start recording; on == 0: stop recording (records stay readable)
extern "C" int ptv2_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (on) {
        for (int k = 0; k < KID_COUNT; ++k) g_seen[k] = 0;
        if (!g_stamp_buf) {  // (outside any capture: bench.py enables the timer between steps)
            if (hipMalloc((void **)&g_stamp_buf, sizeof(unsigned long long) * 2 * STAMP_CAP) != hipSuccess) g_stamp_buf = nullptr;
            int dev = 0, khz = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0)
                g_stamp_us_per_tick = 1e3 / (double)khz;
        }
        g_stamp_n = 0;
        for (int k = 0; k < KID_COUNT; ++k) { g_stamp_recs[k].clear(); g_stamp_pending[k].clear(); }
        for (int k = 0; k < KID_COUNT; ++k) {
            for (auto &r : g_recs[k]) { g_free.push_back(r.a); g_free.push_back(r.b); }
            g_recs[k].clear();
            for (auto e : g_pending[k]) g_free.push_back(e);
            g_pending[k].clear();
        }
    }
    g_on = on ? 1 : 0;
    return PTV2_OK;
}

extern "C" int ptv2_profile_kernel_count(void) { return KID_COUNT; }

// microseconds of one EMPTY stamp bracket [stamp, stamp] inside a captured sequence, median of `reps` (bench.py subtracts it
// from the stamp-bracketed averages, as it subtracts the empty HIP-event bracket from the event-bracketed ones)
extern "C" double ptv2_profile_empty_stamp_us(void *stream, int reps) {
    (void)stream;  // (the null stream cannot capture: a stream of its own)
    if (!g_stamp_buf || reps < 1 || reps > 1024) return -1.0;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -1.0;
    unsigned long long *tmp = nullptr;
    if (hipMalloc((void **)&tmp, sizeof(unsigned long long) * 2 * reps) != hipSuccess) { (void)hipStreamDestroy(st); return -1.0; }
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    double out = -1.0;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess) {
        for (int i = 0; i < 2 * reps; ++i) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, st, tmp + i);
        if (hipStreamEndCapture(st, &g) == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess &&
            hipGraphLaunch(ge, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
            std::vector<unsigned long long> host((size_t)2 * reps);
            if (hipMemcpy(host.data(), tmp, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost) == hipSuccess) {
                std::vector<double> d;
                for (int i = 0; i < reps; ++i) d.push_back(g_stamp_us_per_tick * (double)(host[2 * i + 1] - host[2 * i]));
                std::sort(d.begin(), d.end());
                out = d[d.size() / 2];
            }
        }
    }
    (void)hipGetLastError();
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    (void)hipFree(tmp);
    (void)hipStreamDestroy(st);
    return out;
}

// Synchronises on the recorded events.  Returns PTV2_OK and fills name (>= 64 bytes), total microseconds,
// number of launches and mean algorithmic bytes per launch of kernel id `kid`.
extern "C" int ptv2_profile_read(int kid, char *name, double *total_us, long long *launches, double *bytes_per_launch) {
    if (kid < 0 || kid >= KID_COUNT) return PTV2_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    double us = 0.0, bytes = 0.0;
    for (auto &r : g_recs[kid]) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.b);
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) us += 1e3 * (double)ms;
        bytes += r.bytes;
    }
    if (!g_stamp_recs[kid].empty() && g_stamp_buf) {
        std::vector<unsigned long long> host((size_t)2 * g_stamp_n);
        (void)hipDeviceSynchronize();
        if (hipMemcpy(host.data(), g_stamp_buf, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost) == hipSuccess)
            for (auto &r : g_stamp_recs[kid]) {
                us += g_stamp_us_per_tick * (double)(host[2 * r.slot + 1] - host[2 * r.slot]);
                bytes += r.bytes;
            }
    }
    const size_t count = g_recs[kid].size() + g_stamp_recs[kid].size();
    int i = 0;
    for (; kNames[kid][i] && i < 63; ++i) name[i] = kNames[kid][i];
    name[i] = 0;
    *total_us = us;
    *launches = (long long)count;
    *bytes_per_launch = count == 0 ? 0.0 : bytes / (double)count;
    return PTV2_OK;
}
