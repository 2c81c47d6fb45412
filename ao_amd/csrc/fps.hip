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
#include <algorithm>
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
// W workgroups per cloud, every point of the cloud resident in VGPRs (RC = 2, 4 or 8 per lane), one exchange per
// sample: each workgroup publishes its best candidate as ONE 8-byte granule
//     [ float bits of tmp : 32 | tie key (bitrev(t) << 11 | j) inverted : 21 | sample number mod 2048 : 11 ]
// with a device-scope (sc1) store into slots[cloud][parity][w]; lanes 0..W-1 of wave 0 poll the W granules of the
// parity with relaxed device-scope loads until all carry the current sample number, reduce them with a wave max
// and broadcast the winner through LDS.  A granule is data and flag in one naturally aligned store, so no fence
// is needed (MI355X_MICROARCH.md, hand-off price list, row handoff-1to1); two parities suffice because a
// workgroup can only publish sample i+2 after it has consumed every granule of sample i+1, which nobody
// publishes before consuming sample i.  All b*W workgroups must be co-resident: the launcher keeps b*W <= 256
// (one 1024-thread workgroup per CU) and bounds every spin.
constexpr int COOP_TAG_BITS = 11;
constexpr unsigned COOP_TAG_MASK = (1u << COOP_TAG_BITS) - 1u;

// tie part of the granule of point `rel` (21 bits, larger wins): fixed per point, formed once
__device__ __forceinline__ unsigned coop_tie(int rel, int B, int logB) {
    const unsigned t = (unsigned)rel & (unsigned)(B - 1);
    const unsigned j = (unsigned)rel >> logB;                         // < 2^11 (n_b < 2^21 checked by the launcher)
    const unsigned tr = logB ? (__brev(t) >> (32 - logB)) : 0u;       // < 2^10
    return (~((tr << 11) | j)) & ((1u << 21) - 1u);
}

// wave-wide maximum of an unsigned 32-bit value in every lane... of lane 63: the classic DPP ladder (row_shr 1, 2, 3 folded
// into 1 + 2, row_shr 4 / 8 via bank masks, row_bcast15, row_bcast31); the maximum ends up in lane 63 and is read back with
// v_readlane.  Six v_max_u32_dpp instead of six ds_bpermute round trips.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    int x = (int)v;
    x = (int)max((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));  // row_shr:1
    x = (int)max((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));  // row_shr:2
    x = (int)max((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xe, false));  // row_shr:4
    x = (int)max((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xc, false));  // row_shr:8
    x = (int)max((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));  // row_bcast:15
    x = (int)max((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane(x, 63);
}
// maximum of (hi, lo) pairs compared as one 64-bit key, wave-uniform result: max of hi, then max of lo among the holders
__device__ __forceinline__ unsigned long long wave_max_pair(unsigned hi, unsigned lo) {
    const unsigned mh = wave_max_u32(hi);
    const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}

// The winner's COORDINATES travel with its key: a workgroup publishes four granules per sample -- key, and x / y / z each
// as [float bits : 32 | sample tag : 32] -- and wavefront 0 polls the 4 W granules of the parity at once (W <= 16), so the
// next sample's distance update starts from registers instead of from a dependent global load of xyz[winner] (an L2 round
// trip on every sample's critical path).
template <int RC>
__global__ __launch_bounds__(FPS_THREADS) void fps_coop_kernel(const float *__restrict__ xyz,
                                                               const int *__restrict__ offset,
                                                               const int *__restrict__ new_offset, float *tmp,
                                                               int *__restrict__ idx, int B, int logB, int W,
                                                               unsigned long long *slots, int *error_flag) {
    __shared__ unsigned long long s_key[FPS_THREADS / WAVE];
    __shared__ float s_cand[FPS_THREADS / WAVE][3];
    __shared__ unsigned long long s_win;
    __shared__ float s_wxyz[3];
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
    const int chunk = ((cnt + W - 1) / W + FPS_THREADS - 1) / FPS_THREADS * FPS_THREADS;  // <= RC * 1024
    const int base = wg * chunk;
    float px[RC], py[RC], pz[RC], pt[RC];
    unsigned tie[RC];  // low word of the point's key without the sample tag (0 for an unused slot: never wins)
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const int rel = base + tid + i * FPS_THREADS;
        const bool ok = rel < cnt && tid + i * FPS_THREADS < chunk;
        const int k = start_n + (ok ? rel : 0);
        px[i] = xyz[3 * k]; py[i] = xyz[3 * k + 1]; pz[i] = xyz[3 * k + 2];
        pt[i] = ok ? tmp[k] : -1.0f;
        tie[i] = ok ? (coop_tie(rel, B, logB) << COOP_TAG_BITS) : 0u;
    }
    unsigned long long *my_slots = slots + (size_t)cloud * 2 * 4 * W;  // [parity][key, x, y, z][W]
    float x1 = xyz[3 * start_n], y1 = xyz[3 * start_n + 1], z1 = xyz[3 * start_n + 2];
    for (int j = start_m + 1; j < end_m; ++j) {
        const unsigned tag = (unsigned)((j - start_m) % (int)COOP_TAG_MASK) + 1u;  // 1..2047, never the memset value 0
        unsigned bh = 0u, bl = 0u;  // "no candidate": dist bits 0, tie 0
        float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            if (pt[i] >= 0.0f) {
                const float d = ref_d2(px[i], py[i], pz[i], x1, y1, z1);
                const float d2 = fminf(d, pt[i]);
                pt[i] = d2;
                const unsigned h = __float_as_uint(d2);
                const bool take = h > bh || (h == bh && tie[i] > bl);
                bh = take ? h : bh;
                bl = take ? tie[i] : bl;
                bx = take ? px[i] : bx; by = take ? py[i] : by; bz = take ? pz[i] : bz;
            }
        }
        const unsigned long long wbest = wave_max_pair(bh, bl);
        // the holder: keys of real points are unique; among lanes without one (all-empty wavefront) the lowest
        if (lane == __builtin_ctzll(__ballot((((unsigned long long)bh << 32) | bl) == wbest))) {
            s_key[wid] = wbest | tag;
            s_cand[wid][0] = bx; s_cand[wid][1] = by; s_cand[wid][2] = bz;
        }
        __syncthreads();
        if (wid == 0) {
            const unsigned long long mine = lane < FPS_THREADS / WAVE ? s_key[lane] : 0ull;
            const unsigned long long v = wave_max_pair((unsigned)(mine >> 32), (unsigned)mine);  // (all tags equal: the maximum keeps it)
            const int ww = __builtin_ctzll(__ballot(mine == v));  // the wavefront that holds the workgroup's best
            unsigned long long *par = my_slots + (size_t)(j & 1) * 4 * W;
            if (lane < 4) {
                const unsigned long long g = lane == 0 ? v : (((unsigned long long)__float_as_uint(s_cand[ww][lane - 1]) << 32) | tag);
                __hip_atomic_store(par + (size_t)lane * W + wg, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned long long got = (unsigned long long)tag;
            if (lane < 4 * W) {
                int spins = 0;
                for (;;) {
                    got = __hip_atomic_load(par + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (((unsigned)got & COOP_TAG_MASK) == tag) break;
                    if (++spins > (1 << 22)) { *error_flag = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const unsigned long long keyv = lane < W ? got : 0ull;
            const unsigned long long w = wave_max_pair((unsigned)(keyv >> 32), (unsigned)keyv);
            const int wl = __builtin_ctzll(__ballot(keyv == w));  // the workgroup that holds the cloud's best
            const unsigned ghi = (unsigned)(got >> 32);
            const float wx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)ghi, W + wl));
            const float wy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)ghi, 2 * W + wl));
            const float wz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)ghi, 3 * W + wl));
            if (lane == 0) { s_win = w; s_wxyz[0] = wx; s_wxyz[1] = wy; s_wxyz[2] = wz; }
        }
        __syncthreads();
        x1 = s_wxyz[0]; y1 = s_wxyz[1]; z1 = s_wxyz[2];
        if (wg == 0 && tid == 0) {
            const unsigned long long win = s_win;
            const unsigned tie_w = (~(unsigned)(win >> COOP_TAG_BITS)) & ((1u << 21) - 1u);
            const unsigned jj = tie_w & ((1u << 11) - 1u);
            const unsigned tr = tie_w >> 11;
            const unsigned t = logB ? (__brev(tr) >> (32 - logB)) : 0u;
            idx[j] = start_n + (int)(jj * (unsigned)B + t);
        }
    }
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const int rel = base + tid + i * FPS_THREADS;
        if (rel < cnt && tid + i * FPS_THREADS < chunk) tmp[start_n + rel] = pt[i];
    }
}

}  // namespace

extern "C" size_t farthest_point_sampling_hip_workspace_bytes(int b, int n_total) {
    (void)n_total;
    return sizeof(unsigned long long) * 2 * 4 * 256 * (size_t)(b > 0 ? b : 1) + 256;  // granule slots (key, x, y, z) + error flag
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
    // cooperative variant: every cloud split over W <= 16 workgroups, all points in registers (2 .. 16 per lane).  (More
    // workgroups with fewer points each measured no faster -- 40 x 2 points per lane: 2.58 us per sample against 2.46 with
    // 10 x 8 -- the sample's critical path is the exchange and the hop of the winner's coordinates, not the distance update.)
    const int Wcap = std::min(16, 256 / b);  // 4 W granules are polled by the 64 lanes of one wavefront
    int RC = 2;
    while (RC < 16 && (n_max + RC * FPS_THREADS - 1) / (RC * FPS_THREADS) > Wcap) RC *= 2;
    const int W = (n_max + RC * FPS_THREADS - 1) / (RC * FPS_THREADS);
    const bool coop = W >= 2 && W <= Wcap && n_max < (1 << 21) && workspace &&
                      workspace_bytes >= farthest_point_sampling_hip_workspace_bytes(b, n_total);
    if (coop) {
        unsigned long long *slots = (unsigned long long *)workspace;
        int *err = (int *)((char *)workspace + sizeof(unsigned long long) * 2 * 4 * 256 * (size_t)b);
        (void)hipMemsetAsync(workspace, 0, sizeof(unsigned long long) * 2 * 4 * 256 * (size_t)b + sizeof(int), st);
        if (RC == 2)
            hipLaunchKernelGGL(fps_coop_kernel<2>, dim3(b * W), dim3(FPS_THREADS), 0, st, xyz, offset, new_offset, tmp, idx, B, logB, W, slots, err);
        else if (RC == 4)
            hipLaunchKernelGGL(fps_coop_kernel<4>, dim3(b * W), dim3(FPS_THREADS), 0, st, xyz, offset, new_offset, tmp, idx, B, logB, W, slots, err);
        else if (RC == 16)
            hipLaunchKernelGGL(fps_coop_kernel<16>, dim3(b * W), dim3(FPS_THREADS), 0, st, xyz, offset, new_offset, tmp, idx, B, logB, W, slots, err);
        else
            hipLaunchKernelGGL(fps_coop_kernel<8>, dim3(b * W), dim3(FPS_THREADS), 0, st, xyz, offset, new_offset, tmp, idx, B, logB, W, slots, err);
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    hipLaunchKernelGGL(fps_kernel, dim3(b), dim3(FPS_THREADS), 0, st, xyz, offset, new_offset, tmp, idx, B, logB);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
