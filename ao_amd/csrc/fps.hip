// ao_amd/csrc/fps.hip -- farthest point sampling for gfx950, bit-exact vs. the reference
// (libs/pointops/src/sampling/sampling_cuda_kernel.cu:14-129).
//
// The reference runs ONE block of B = opt_n_threads(n_max) threads per cloud and re-reads the
// whole cloud (12 B xyz + 8 B tmp per point) from memory on each of the m-1 dependent sweeps.
// Here a 1024-thread workgroup per cloud keeps the first 12 points of every thread (12 K points,
// 192 KiB of x/y/z/tmp) in VGPRs for the whole run and only streams the remainder of larger
// clouds; the arg-max is a packed 64-bit key reduced with wave shuffles + one LDS hop, one
// barrier per sample.
//
// Tie rule.  The reference's per-thread strict `>` scan (:49-59) followed by the shared-memory
// tree that keeps the lower slot on ties (:5-10, :63-123) selects, among points with equal
// maximal tmp, the one with the smallest (bit-reverse_{log2 B}((k - start) mod B), k)
// (tests/test_oracle_ops.py::test_fps_golden_and_tie_rule).  The key
//     [ float bits of tmp (>= 0) | ~( bitrev(t) << 21 | j ) ],  t = (k-start) mod B, j = (k-start) / B
// makes that rule a plain unsigned max, independent of how points are spread over lanes.
#include <cmath>

#include "common.h"

namespace {

constexpr int FPS_THREADS = 1024;
constexpr int FPS_RC = 12;  // register-cached points per thread

__device__ __forceinline__ unsigned long long fps_key(float d, int rel, int B, int logB) {
    unsigned t = (unsigned)rel & (unsigned)(B - 1);
    unsigned j = (unsigned)rel >> logB;
    unsigned tr = logB ? (__brev(t) >> (32 - logB)) : 0u;
    unsigned lo = ~((tr << 21) | j);
    return ((unsigned long long)__float_as_uint(d) << 32) | lo;
}

__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float *__restrict__ xyz,
                                                          const int *__restrict__ offset,
                                                          const int *__restrict__ new_offset, float *tmp,
                                                          int *__restrict__ idx, int B, int logB) {
    __shared__ unsigned long long s_key[2][FPS_THREADS / WAVE];
    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int start_n = bid == 0 ? 0 : offset[bid - 1];
    const int end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1];
    const int end_m = new_offset[bid];
    const int cnt = end_n - start_n;
    if (end_m <= start_m) return;
    if (tid == 0) idx[start_m] = start_n;  // :39
    if (cnt <= 0) return;

    float px[FPS_RC], py[FPS_RC], pz[FPS_RC], pt[FPS_RC];
#pragma unroll
    for (int i = 0; i < FPS_RC; ++i) {
        int rel = tid + i * FPS_THREADS;
        bool ok = rel < cnt;
        int k = start_n + (ok ? rel : 0);
        px[i] = xyz[3 * k];
        py[i] = xyz[3 * k + 1];
        pz[i] = xyz[3 * k + 2];
        pt[i] = ok ? tmp[k] : -1.0f;  // tmp < 0 marks an unused slot
    }

    int old = start_n;
    for (int j = start_m + 1; j < end_m; ++j) {
        const float x1 = xyz[3 * old], y1 = xyz[3 * old + 1], z1 = xyz[3 * old + 2];
        unsigned long long best = 0ull;
#pragma unroll
        for (int i = 0; i < FPS_RC; ++i) {
            if (pt[i] >= 0.0f) {
                float d = ref_d2(px[i], py[i], pz[i], x1, y1, z1);  // (x2-x1)^2.. point minus last sample (:54)
                float d2 = fminf(d, pt[i]);
                pt[i] = d2;
                unsigned long long key = fps_key(d2, tid + i * FPS_THREADS, B, logB);
                best = key > best ? key : best;
            }
        }
        for (int rel = tid + FPS_RC * FPS_THREADS; rel < cnt; rel += FPS_THREADS) {  // clouds > 12 K points
            int k = start_n + rel;
            float d = ref_d2(xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], x1, y1, z1);
            float d2 = fminf(d, tmp[k]);
            tmp[k] = d2;
            unsigned long long key = fps_key(d2, rel, B, logB);
            best = key > best ? key : best;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            unsigned long long other = shfl_xor_u64(best, o);
            best = other > best ? other : best;
        }
        const int buf = j & 1;
        if (lane == 0) s_key[buf][wid] = best;
        __syncthreads();
        unsigned long long win = s_key[buf][0];
#pragma unroll
        for (int w = 1; w < FPS_THREADS / WAVE; ++w) {
            unsigned long long o = s_key[buf][w];
            win = o > win ? o : win;
        }
        const unsigned x = ~(unsigned)win;
        const unsigned jj = x & ((1u << 21) - 1u);
        const unsigned tr = x >> 21;
        const unsigned t = logB ? (__brev(tr) >> (32 - logB)) : 0u;
        old = start_n + (int)(jj * (unsigned)B + t);
        if (tid == 0) idx[j] = old;
    }
#pragma unroll
    for (int i = 0; i < FPS_RC; ++i) {  // leave tmp as the reference does: min squared distance to the sample set
        int rel = tid + i * FPS_THREADS;
        if (rel < cnt) tmp[start_n + rel] = pt[i];
    }
}


// ------------------------------------------------------------- cooperative FPS --
// W workgroups per cloud, every point of the cloud resident in VGPRs (<= COOP_RC per lane), one exchange per
// sample: each workgroup publishes its best candidate as ONE 8-byte granule
//     [ float bits of tmp : 32 | tie key (bitrev(t) << 11 | j) inverted : 21 | sample number mod 2048 : 11 ]
// with a device-scope (sc1) store into slots[cloud][parity][w]; lanes 0..W-1 of wave 0 poll the W granules of the
// parity with relaxed device-scope loads until all carry the current sample number, reduce them with a wave max
// and broadcast the winner through LDS.  A granule is data and flag in one naturally aligned store, so no fence
// is needed (MI355X_MICROARCH.md, hand-off price list, row handoff-1to1); two parities suffice because a
// workgroup can only publish sample i+2 after it has consumed every granule of sample i+1, which nobody
// publishes before consuming sample i.  All b*W workgroups must be co-resident: the launcher keeps b*W <= 256
// (one 1024-thread workgroup per CU) and bounds every spin.
constexpr int COOP_RC = 8;          // points per lane  -> 8192 points per workgroup
constexpr int COOP_TAG_BITS = 11;
constexpr unsigned COOP_TAG_MASK = (1u << COOP_TAG_BITS) - 1u;

__device__ __forceinline__ unsigned long long coop_key(float d, int rel, int B, int logB, unsigned tag) {
    const unsigned t = (unsigned)rel & (unsigned)(B - 1);
    const unsigned j = (unsigned)rel >> logB;                         // < 2^11 (n_b < 2^21 checked by the launcher)
    const unsigned tr = logB ? (__brev(t) >> (32 - logB)) : 0u;       // < 2^10
    const unsigned tie = (~((tr << 11) | j)) & ((1u << 21) - 1u);
    return ((unsigned long long)__float_as_uint(d) << 32) | ((unsigned long long)tie << COOP_TAG_BITS) | tag;
}

__global__ __launch_bounds__(FPS_THREADS) void fps_coop_kernel(const float *__restrict__ xyz,
                                                               const int *__restrict__ offset,
                                                               const int *__restrict__ new_offset, float *tmp,
                                                               int *__restrict__ idx, int B, int logB, int W,
                                                               unsigned long long *slots, int *error_flag) {
    __shared__ unsigned long long s_key[FPS_THREADS / WAVE];
    __shared__ unsigned long long s_win;
    const int cloud = blockIdx.x / W, wg = blockIdx.x - cloud * W;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int start_n = cloud == 0 ? 0 : offset[cloud - 1];
    const int end_n = offset[cloud];
    const int start_m = cloud == 0 ? 0 : new_offset[cloud - 1];
    const int end_m = new_offset[cloud];
    const int cnt = end_n - start_n;
    if (end_m <= start_m) return;
    if (wg == 0 && tid == 0) idx[start_m] = start_n;
    if (cnt <= 0) return;
    const int chunk = ((cnt + W - 1) / W + FPS_THREADS - 1) / FPS_THREADS * FPS_THREADS;  // <= COOP_RC * 1024
    const int base = wg * chunk;
    float px[COOP_RC], py[COOP_RC], pz[COOP_RC], pt[COOP_RC];
#pragma unroll
    for (int i = 0; i < COOP_RC; ++i) {
        const int rel = base + tid + i * FPS_THREADS;
        const bool ok = rel < cnt && tid + i * FPS_THREADS < chunk;
        const int k = start_n + (ok ? rel : 0);
        px[i] = xyz[3 * k]; py[i] = xyz[3 * k + 1]; pz[i] = xyz[3 * k + 2];
        pt[i] = ok ? tmp[k] : -1.0f;
    }
    unsigned long long *my_slots = slots + (size_t)cloud * 2 * W;
    int old = start_n;
    for (int j = start_m + 1; j < end_m; ++j) {
        const unsigned tag = (unsigned)((j - start_m) % (int)COOP_TAG_MASK) + 1u;  // 1..2047, never the memset value 0
        const float x1 = xyz[3 * old], y1 = xyz[3 * old + 1], z1 = xyz[3 * old + 2];
        unsigned long long best = (unsigned long long)tag;  // "no candidate": dist bits 0, tie 0
#pragma unroll
        for (int i = 0; i < COOP_RC; ++i) {
            if (pt[i] >= 0.0f) {
                const float d = ref_d2(px[i], py[i], pz[i], x1, y1, z1);
                const float d2 = fminf(d, pt[i]);
                pt[i] = d2;
                const unsigned long long key = coop_key(d2, base + tid + i * FPS_THREADS, B, logB, tag);
                best = key > best ? key : best;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned long long other = shfl_xor_u64(best, o);
            best = other > best ? other : best;
        }
        if (lane == 0) s_key[wid] = best;
        __syncthreads();
        if (wid == 0) {
            unsigned long long v = lane < FPS_THREADS / WAVE ? s_key[lane] : 0ull;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) {
                const unsigned long long other = shfl_xor_u64(v, o);
                v = other > v ? other : v;
            }
            unsigned long long *par = my_slots + (size_t)(j & 1) * W;
            if (lane == 0) __hip_atomic_store(par + wg, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long got = (unsigned long long)tag;
            if (lane < W) {
                int spins = 0;
                for (;;) {
                    got = __hip_atomic_load(par + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (((unsigned)got & COOP_TAG_MASK) == tag) break;
                    if (++spins > (1 << 22)) { *error_flag = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const unsigned long long other = shfl_xor_u64(got, o);
                got = other > got ? other : got;
            }
            if (lane == 0) s_win = got;
        }
        __syncthreads();
        const unsigned long long win = s_win;
        const unsigned tie = (~(unsigned)(win >> COOP_TAG_BITS)) & ((1u << 21) - 1u);
        const unsigned jj = tie & ((1u << 11) - 1u);
        const unsigned tr = tie >> 11;
        const unsigned t = logB ? (__brev(tr) >> (32 - logB)) : 0u;
        old = start_n + (int)(jj * (unsigned)B + t);
        if (wg == 0 && tid == 0) idx[j] = old;
    }
#pragma unroll
    for (int i = 0; i < COOP_RC; ++i) {
        const int rel = base + tid + i * FPS_THREADS;
        if (rel < cnt && tid + i * FPS_THREADS < chunk) tmp[start_n + rel] = pt[i];
    }
}

}  // namespace

extern "C" size_t farthest_point_sampling_hip_workspace_bytes(int b, int n_total) {
    (void)n_total;
    return sizeof(unsigned long long) * 2 * 256 * (size_t)(b > 0 ? b : 1) + 256;  // granule slots + error flag
}

extern "C" int farthest_point_sampling_hip_launcher(int b, int n_max, const float *xyz, const int *offset,
                                                    const int *new_offset, float *tmp, int *idx, int n_total,
                                                    int m_total, void *workspace, size_t workspace_bytes,
                                                    void *stream) {
    (void)n_total;
    if (b < 1 || n_max < 1) return PTV2_ERR_ARG;
    if (m_total <= 0) return PTV2_OK;
    if (!xyz || !offset || !new_offset || !tmp || !idx) return PTV2_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // cuda_utils.h:11-14 opt_n_threads(): 2^floor(log2 n_max) clamped to [1, 1024], in double as there
    const int pow_2 = (int)(std::log((double)n_max) / std::log(2.0));
    int B = 1 << pow_2;
    B = B > 1024 ? 1024 : (B < 1 ? 1 : B);
    int logB = 0;
    while ((1 << logB) < B) ++logB;
    // cooperative variant: every cloud split over W workgroups, all points in registers
    const int W = (n_max + COOP_RC * FPS_THREADS - 1) / (COOP_RC * FPS_THREADS);
    const bool coop = W >= 2 && W <= 64 && (long long)b * W <= 256 && n_max < (1 << 21) && workspace &&
                      workspace_bytes >= farthest_point_sampling_hip_workspace_bytes(b, n_total);
    if (coop) {
        unsigned long long *slots = (unsigned long long *)workspace;
        int *err = (int *)((char *)workspace + sizeof(unsigned long long) * 2 * 256 * (size_t)b);
        (void)hipMemsetAsync(workspace, 0, sizeof(unsigned long long) * 2 * 256 * (size_t)b + sizeof(int), st);
        hipLaunchKernelGGL(fps_coop_kernel, dim3(b * W), dim3(FPS_THREADS), 0, st, xyz, offset, new_offset, tmp, idx, B, logB,
                           W, slots, err);
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    hipLaunchKernelGGL(fps_kernel, dim3(b), dim3(FPS_THREADS), 0, st, xyz, offset, new_offset, tmp, idx, B, logB);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
