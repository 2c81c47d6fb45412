// ao_amd/csrc/common.h -- shared device/host helpers for libptv2_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptv2_hip.h"

#define WAVE 64

#define PTV2_CHECK_LAUNCH()                         \
    do {                                            \
        if (hipGetLastError() != hipSuccess) return PTV2_ERR_LAUNCH; \
    } while (0)

static inline int divup(long long a, long long b) { return (int)((a + b - 1) / b); }

// Optional per-kernel timing for bench.py's roofline leg (abi.hip): when enabled through
// ptv2_profile_enable(1), launchers bracket their main kernel with HIP events on the launch stream.
enum PtvKernelId {
    KID_KNN_QUERY = 0, KID_LOGITS_FWD, KID_SOFTMAX_ROWS, KID_AGG_TILE, KID_PEB_FWD, KID_PEB_BWD, KID_BWD_TILE,
    KID_BWD_ROWS, KID_BWD_GV, KID_LOGITS_BWD_ROWS, KID_LOGITS_BWD_GATHER, KID_LOGITS_BWD_PARAMS, KID_WGRAD,
    KID_BN_STATS, KID_BN_APPLY, KID_BN_BWD_REDUCE, KID_BN_BWD_APPLY, KID_SKINNY_FWD, KID_SKINNY_BWD, KID_ROWS_GEMM, /* + 0..7: (BN 48|64) x (W (n,k)|(k,n)) x (KC 32|64) */ KID_BWD_POINT = KID_ROWS_GEMM + 8,
    /* + 0..4 for G = 6, 12, 24, 48, 64 */ KID_WGRAD_LDS = KID_BWD_POINT + 5 /* the LDS-staged fp32 weight gradient */,
    KID_WGRAD_GROUPED /* the grouped projection's weight gradient on the vector ALUs */,
    KID_FWD_POINT /* softmax + aggregation + grouped projection of the forward in one launch (gva_fwd_point.hip) */,
    KID_LOGITS_BWD_FUSED /* + 0..4 for G = 6, 12, 24, 48, 64: rows + parameter gradients of the logits stage in one launch */,
    KID_BN_BWD_APPLY_RES = KID_LOGITS_BWD_FUSED + 5 /* bn_bwd_apply_residual_kernel */,
    KID_BN_BWD_FINAPPLY /* bn_bwd_finapply_kernel<...>: record sum + apply in one launch */,
    KID_FWD_TILE /* + 0..3 for G = 12, 24, 48, 64: softmax + aggregation + grouped projection per 16-point tile (gva_fwd_tile.hip) */,
    KID_WGRAD_TILE = KID_FWD_TILE + 4 /* the grouped projection's weight gradient with A recomputed (gva_wgrad_tile.hip) */,
    KID_BWD_TILE_K /* + 0..3 for G = 12, 24, 48, 64: the deep levels' attention backward per tile of points (gva_bwd_tile.hip) */,
    KID_COUNT = KID_BWD_TILE_K + 4
};
extern "C" int ptv2_profile_is_on(void);
int ptv2_profile_wants(int kid);
void ptv2_profile_begin(int kid, hipStream_t st);
void ptv2_profile_end(int kid, hipStream_t st, double algorithmic_bytes);
struct PtvScopedTimer {
    int kid; hipStream_t st; double bytes; bool on;
    PtvScopedTimer(int k, hipStream_t s, double b) : kid(k), st(s), bytes(b), on(ptv2_profile_wants(k) != 0) {
        if (on) ptv2_profile_begin(kid, st);
    }
    ~PtvScopedTimer() { if (on) ptv2_profile_end(kid, st, bytes); }
};

// graph.hip: the launcher body between construction and finish() is captured and issued as one hipGraph launch
enum PtvGraphSlot { GRAPH_MODEL_FWD = 0, GRAPH_MODEL_BWD = 1, GRAPH_MODEL_FWD_PREFIX = 2, GRAPH_MODEL_FWD_REST = 3, GRAPH_SLOTS = 4 };
struct PtvGraphScope {
    void *st, *cap; int which, ring_entry; bool active; long long t0;
    PtvGraphScope(void *stream, int which, bool allow = true);
    void *stream() const { return cap; }  // where the body must enqueue: the capture stream, or the caller's when declined
    int finish(int rc);   // returns rc, or PTV2_ERR_LAUNCH when the graph could not be built / launched
    ~PtvGraphScope();     // an unfinished scope (early return) closes and discards its capture
    PtvGraphScope(const PtvGraphScope &) = delete;
    PtvGraphScope &operator=(const PtvGraphScope &) = delete;
};
int ptv2_graph_capturing(void);   // this thread is inside an active scope
int ptv2_profile_stamps(void);    // the kernel timer brackets with device time stamps (legal inside a capture), not HIP events
// zero-fill as a kernel launch (a kernel node like every other launch of a captured sequence; bytes % 4 == 0)
int ptv2_zero_async(void *p, size_t bytes, hipStream_t st);

// dense.hip: the weight-gradient launches of a model backward, filed where they are called and run by one launch at its end
void ptv2_wgrad_defer_begin(void *arena, size_t bytes);   // arena: job table + operands that must outlive their Block + records
bool ptv2_wgrad_defer_active();
float *ptv2_wgrad_defer_alloc(size_t floats);             // NULL: not deferring, or no room (the caller launches at once)
void ptv2_wgrad_defer_arm(bool on);                       // the next weight-gradient call may be filed (its operands are kept)
void ptv2_wgrad_defer_arm_rs(bool on);                    // ... the row-scaled strided form inside the attention backward
bool ptv2_wgrad_defer_armed_rs();
int ptv2_wgrad_defer_flush(void *stream);                 // run what has been filed
void ptv2_wgrad_defer_end();
size_t ptv2_wgrad_defer_table_bytes();

// Matmul operand precision of the calling thread's launches (abi.hip): 0 = fp32 MFMA (V_MFMA_F32_16X16X4_F32, exact fp32),
// 1 = bf16 MFMA (V_MFMA_F32_16X16X32_BF16: operands rounded to bf16 on their way into the matrix core, fp32
// accumulation) -- what torch.autocast(dtype=bfloat16) does to the nn.Linear layers of the reference
// (pointcept/engines/train_sam_pp2s.py:178-180).  Set by the Block / model launchers from their `matmul_bf16` field for the
// duration of the call; the row GEMM and the weight-gradient launchers read it.
int ptv2_matmul_bf16(void);
void ptv2_set_matmul_bf16(int on);
struct PtvMatmulScope {
    int prev;
    explicit PtvMatmulScope(int on) : prev(ptv2_matmul_bf16()) { ptv2_set_matmul_bf16(on); }
    ~PtvMatmulScope() { ptv2_set_matmul_bf16(prev); }
};
typedef __bf16 ptv2_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ ptv2_bf16x8 ptv2_pack_bf16(float4 lo, float4 hi) {  // v_cvt_pk_bf16_f32 x 4 (round to nearest even)
    ptv2_bf16x8 r;
    r[0] = (__bf16)lo.x; r[1] = (__bf16)lo.y; r[2] = (__bf16)lo.z; r[3] = (__bf16)lo.w;
    r[4] = (__bf16)hi.x; r[5] = (__bf16)hi.y; r[6] = (__bf16)hi.z; r[7] = (__bf16)hi.w;
    return r;
}

// Target of masked loads: `*(ok ? p : (const T *)ptv2_zero_pad)` keeps a kernel's loads UNCONDITIONAL and still reads 0 in the
// masked lanes.  A load under a divergent condition (`ok ? *p : 0`) cannot be speculated by the compiler: it becomes a basic
// block of its own, and the wait-count insertion then drains the whole memory queue (s_waitcnt vmcnt(0)) at the join, i.e.
// one exposed memory round trip per such load (tools/isa_waits.py counts them).  Zero-initialised, never written;
// offsets up to 1024 floats are valid.
static __device__ __attribute__((aligned(16), unused)) float ptv2_zero_pad[1024];
template <class T>
__device__ __forceinline__ T ptv2_ld_or_zero(const T *p, bool ok) {
    return *(ok ? p : (const T *)ptv2_zero_pad);
}

// Zeroed per-stream device counters for kernels that reduce their own per-block partial sums (abi.hip).
#define PTV2_NUM_COUNTERS 64
enum PtvCounterSlot { CNT_LOGITS_FWD = 0, CNT_BP2, CNT_LOGITS_BWD_ROWS, CNT_CE, CNT_BN_TILES = 16 /* .. + 31 */ };
unsigned *ptv2_stream_counters(hipStream_t st);

// Squared distance with the rounding sequence pinned (see oracle/pointops_oracle.c REF_D2):
// fma(dz,dz, fma(dx,dx, dy*dy)), q - p per component.  The only add that could be
// contracted is already inside an explicit fma, so -ffp-contract cannot change it.
__device__ __forceinline__ float ref_d2(float qx, float qy, float qz, float x, float y, float z) {
    float dx = qx - x, dy = qy - y, dz = qz - z;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
}

// First segment i with pt < offset[i] (the reference's get_bt_idx,
// knn_query_cuda_kernel.cu:45-56), as a binary search; clamps to b-1.
__device__ __forceinline__ int seg_of(int pt, const int *__restrict__ offset, int b) {
    int lo = 0, hi = b - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (pt < offset[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Order-preserving float <-> int encoding for atomicMin/atomicMax on floats.
__device__ __forceinline__ int f2ord(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) {
    return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, m, WAVE);
    hi = __shfl_xor(hi, m, WAVE);
    return ((unsigned long long)hi << 32) | lo;
}

// Bijective XCD-aware remap of a 1-D block index (guide T1): blocks that share
// `orig % 8` share an XCD (and its L2); give each such group a contiguous range
// of logical tiles.  Speed only -- never correctness.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = orig & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (orig >> 3);
}
