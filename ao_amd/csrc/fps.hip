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
// W workgroups per cloud (W <= 32), every point of the cloud resident in VGPRs (RC = 2, 4, 8 or 16 per lane), one exchange
// per sample.  A workgroup publishes its best candidate as four 8-byte granules, stored as two 16-byte pairs:
//     key  [ float bits of tmp : 32 | tie key (bitrev(t) << 11 | j) inverted : 21 | sample number mod 2047 + 1 : 11 ]
//     x, y, z of that candidate, each [ float bits : 32 | sample tag : 32 ]
// -- the winner's COORDINATES travel with its key, so the next sample's distance update starts from registers instead of from
// a dependent load of xyz[winner].  Wavefront 0 of every workgroup polls the 2 W pairs of the parity (one 16-byte load per
// lane) until all carry the current sample number, reduces the keys with a wave max and broadcasts the winner through LDS.
// A granule is data and flag in one naturally aligned store, so no fence is needed (MI355X_MICROARCH.md, hand-off price list,
// row handoff-1to1); two parities suffice because a workgroup can only publish sample i+2 after it has consumed every granule
// of sample i+1, which nobody publishes before consuming sample i.  All workgroups of a launch must be co-resident (the
// launcher bounds the grid by the compute-unit count) and every spin is bounded.
// Two exchanges (template LOCAL): through the L2 of ONE XCD that holds all W workgroups of the cloud (teams formed at run
// time, see the kernel), or device scope through memory when the launch does not fit that way (AO_AMD_FPS_LOCAL=0 forces it).
// Measured at 120 k points -> 30 k samples: 1.42-1.49 us per sample through the XCD's L2 (W = 30, 4 points per lane),
// 2.37 through memory; round 3's form (branchy update, 8-byte granules polled by key only) 2.58.  What a sample costs now is
// ~1.25 us of hand-offs that do not shrink with the cloud (two barriers, two wave reductions, the store's and the load's trip
// to L2: 16 k .. 64 k points all take 1.27-1.33 us) plus ~0.05 us per point and lane.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int COOP_TAG_BITS = 11;
constexpr int COOP_TEAMS = 32;       // teams per XCD the control block has room for (LOCAL exchange)
constexpr int COOP_CTRL_INTS = 32 + 16 * COOP_TEAMS;  // tickets[16], next cloud, arrived, ..., cloud_of[16][COOP_TEAMS]
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

template <int RC, bool LOCAL>
__global__ __launch_bounds__(FPS_THREADS) void fps_coop_kernel(const float *__restrict__ xyz,
                                                               const int *__restrict__ offset,
                                                               const int *__restrict__ new_offset, float *tmp,
                                                               int *__restrict__ idx, int B, int logB, int W,
                                                               unsigned long long *slots, int *error_flag, int nclouds, int *ctrl) {
    __shared__ unsigned long long s_key[FPS_THREADS / WAVE];
    __shared__ float s_cand[FPS_THREADS / WAVE][3];
    __shared__ unsigned long long s_win;
    __shared__ float s_wxyz[3];
    // LOCAL: all W workgroups of a cloud on ONE XCD, whose L2 then carries the exchange: a granule is stored at workgroup
    // scope (sc0: through the CU's write-through L1 into the XCD's L2, no write-through to memory) and polled with a
    // device-scope load (sc1: misses the reader's L1, hits that L2).  Device-scope stores -- what workgroups on different
    // XCDs need -- go to memory before anyone can see them: 2.41 us per sample against 1.80 at 120 k points.
    // Which XCD a workgroup lands on is the dispatcher's business (round-robin over the XCDs when they all have room:
    // tools/probes/xcc_probe.hip -- but workgroup 112 of a 120-workgroup launch was seen on XCD 7), so the teams form at run
    // time: a workgroup reads its XCC_ID, draws a ticket of that XCD (ticket / W = its team there, ticket % W = its rank), the
    // one that completes a team claims the next cloud for it and tells the others.  Teams that never fill up (the launch has
    // 8 (W - 1) + 1 - W workgroups more than the clouds need: one partial team per XCD cannot starve a cloud) leave once every workgroup of
    // the launch has drawn its ticket.
    int cloud, wg;
    if (LOCAL) {
        __shared__ int s_cloud, s_rank;
        if (threadIdx.x == 0) {
            int *tickets = ctrl, *next_cloud = ctrl + 16, *arrived = ctrl + 17, *cloud_of = ctrl + 32;  // cloud_of[16][COOP_TEAMS]
            const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;  // HW_REG_XCC_ID[3:0]
            const int t = __hip_atomic_fetch_add(tickets + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int team = t / W, rank = t - team * W;
            int mine = -1;
            // every workgroup reports in exactly once; the LAST one knows how many clouds found a team.  Fewer than nclouds
            // (the dispatcher spread the workgroups so unevenly that complete teams were over the cap, or partial teams ate the
            // slack) would leave those clouds' indices unwritten without any spin timing out: fail loudly instead
            auto report_in = [&]() {
                const int before = __hip_atomic_fetch_add(arrived, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1 == (int)gridDim.x &&
                    __hip_atomic_load(next_cloud, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nclouds)
                    *error_flag = 1;
            };
            if (team < COOP_TEAMS) {
                int *slot = cloud_of + xcc * COOP_TEAMS + team;
                if (rank == W - 1) {  // (tickets come in order: ranks 0 .. W-2 of this team are taken, the team is complete)
                    mine = __hip_atomic_fetch_add(next_cloud, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(slot, mine + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    report_in();
                } else {
                    report_in();
                    for (int spins = 0;; ++spins) {
                        if (spins > (1 << 22)) { *error_flag = 1; break; }  // (a launch that is not resident as a whole: fail loudly)
                        int c = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                        if (c == 0 && __hip_atomic_load(arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x)
                            c = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);  // everybody is in: final answer
                        else if (c == 0) { __builtin_amdgcn_s_sleep(8); continue; }
                        mine = c - 1;  // (-1: the team never filled up)
                        break;
                    }
                }
            } else {
                report_in();
            }
            s_cloud = mine;
            s_rank = rank;
        }
        __syncthreads();
        cloud = s_cloud;
        wg = s_rank;
        if (cloud < 0 || cloud >= nclouds) return;
    } else {
        cloud = blockIdx.x / W;
        wg = blockIdx.x - cloud * W;
    }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int STORE_SCOPE = LOCAL ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT;
    __shared__ int s_abort;
    if (tid == 0) s_abort = 0;
    __syncthreads();
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
        pt[i] = ok ? tmp[k] : 0.0f;  // (an unused slot: distance 0, tie 0 -- its key is 0)
        tie[i] = ok ? (coop_tie(rel, B, logB) << COOP_TAG_BITS) : 0u;
    }
    unsigned long long *my_slots = slots + (size_t)cloud * 2 * 4 * W;  // [parity][key, x, y, z][W]
    float x1 = xyz[3 * start_n], y1 = xyz[3 * start_n + 1], z1 = xyz[3 * start_n + 2];
    for (int j = start_m + 1; j < end_m; ++j) {
        const unsigned tag = (unsigned)((j - start_m) % (int)COOP_TAG_MASK) + 1u;  // 1..2047, never the memset value 0
        // branch-free: the candidate is ONE 64-bit key (distance bits : tie), the slot that holds it a small integer.  (The
        // first form tracked (distance, tie, x, y, z) under per-point branches: the compiler turned every point into an
        // exec-mask region -- ~14 scalar / mask instructions around 7 of arithmetic -- and the update, not the exchange,
        // was most of a sample: 1.41 us per sample for ONE workgroup with 8 points per lane and nobody to talk to.)
        unsigned long long best = 0ull;  // "no candidate": an unused slot has pt = 0, tie = 0 -> key 0, never taken
        int bi = 0;
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const float d = ref_d2(px[i], py[i], pz[i], x1, y1, z1);
            float d2;
            asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(pt[i]));  // (fminf would canonicalise pt[i] first: one more op)
            pt[i] = d2;
            const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | tie[i];
            const bool take = key > best;
            best = take ? key : best;
            bi = take ? i : bi;
        }
        const unsigned bh = (unsigned)(best >> 32), bl = (unsigned)best;
        const unsigned long long wbest = wave_max_pair(bh, bl);
        // the holder: keys of real points are unique; among lanes without one (all-empty wavefront) the lowest
        const int holder = __builtin_ctzll(__ballot(best == wbest));
        const int hi_slot = __builtin_amdgcn_readlane(bi, holder);  // (wave-uniform: the selection below is scalar branches)
        float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
        for (int i = 0; i < RC; ++i)
            if (hi_slot == i) { bx = px[i]; by = py[i]; bz = pz[i]; }
        if (lane == holder) {
            s_key[wid] = wbest | tag;
            s_cand[wid][0] = bx; s_cand[wid][1] = by; s_cand[wid][2] = bz;
        }
        __syncthreads();
        if (wid == 0) {
            const bool has = lane < FPS_THREADS / WAVE;
            const unsigned long long mine = has ? s_key[lane] : 0ull;
            // (lane w also fetches wavefront w's candidate now: the winner's coordinates are then a register read away
            // instead of an LDS round trip behind the reduction)
            const float mx = has ? s_cand[lane][0] : 0.f, my = has ? s_cand[lane][1] : 0.f, mz = has ? s_cand[lane][2] : 0.f;
            const unsigned long long v = wave_max_pair((unsigned)(mine >> 32), (unsigned)mine);  // (all tags equal: the maximum keeps it)
            const int ww = __builtin_ctzll(__ballot(mine == v));  // the wavefront that holds the workgroup's best
            const float vx = __uint_as_float((unsigned)__builtin_amdgcn_readlane(__float_as_int(mx), ww));
            const float vy = __uint_as_float((unsigned)__builtin_amdgcn_readlane(__float_as_int(my), ww));
            const float vz = __uint_as_float((unsigned)__builtin_amdgcn_readlane(__float_as_int(mz), ww));
            // the parity's granules as 2 W pairs of 16 bytes: pair w = (key, x) of workgroup w, pair W + w = its (y, z).  One
            // 16-byte store per pair (lanes 0, 1), one 16-byte load per lane and poll round (lanes 0 .. 2 W - 1, W <= 32); every
            // 8-byte granule carries the sample tag, so it does not matter whether the two halves of a pair arrive together
            u32x4 *par = (u32x4 *)(my_slots + (size_t)(j & 1) * 4 * W);
            if (lane < 2) {
                const float c0 = lane == 0 ? vx : vy, c1 = vz;
                u32x4 g;
                if (lane == 0) g = u32x4{(unsigned)v, (unsigned)(v >> 32), tag, __float_as_uint(c0)};
                else g = u32x4{tag, __float_as_uint(c0), tag, __float_as_uint(c1)};
                u32x4 *dst = par + (size_t)lane * W + wg;
                if (LOCAL) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(dst), "v"(g) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(g) : "memory");
            }
            u32x4 got = u32x4{tag, 0u, tag, 0u};
            if (lane < 2 * W) {
                const u32x4 *src = par + lane;
                int spins = 0;
                for (;;) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(src) : "memory");
                    if ((got.x & COOP_TAG_MASK) == tag && (got.z & COOP_TAG_MASK) == tag) break;
                    if (++spins > (1 << 20)) { *error_flag = 1; s_abort = 1; break; }
                    if (!LOCAL) __builtin_amdgcn_s_sleep(1);  // (the XCD's L2 takes the tight loop; memory-side polling backs off)
                }
            }
            const unsigned long long keyv = lane < W ? (((unsigned long long)got.y << 32) | got.x) : 0ull;
            const unsigned long long w = wave_max_pair((unsigned)(keyv >> 32), (unsigned)keyv);
            const int wl = __builtin_ctzll(__ballot(keyv == w));  // the workgroup that holds the cloud's best
            const float wx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)got.w, wl));
            const float wy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)got.y, W + wl));
            const float wz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)got.w, W + wl));
            if (lane == 0) { s_win = w; s_wxyz[0] = wx; s_wxyz[1] = wy; s_wxyz[2] = wz; }
        }
        __syncthreads();
        if (s_abort) return;  // (a granule never came: the launcher's caller raises; do not spin through the remaining samples)
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

// workgroups of the LOCAL kernel the device holds at once (occupancy x compute units), per RC
int coop_capacity(int RC) {
    static int cap[4] = {0, 0, 0, 0};
    const int slot = RC == 2 ? 0 : RC == 4 ? 1 : RC == 8 ? 2 : 3;
    if (!cap[slot]) {
        const void *fn = RC == 2 ? (const void *)fps_coop_kernel<2, true> : RC == 4 ? (const void *)fps_coop_kernel<4, true>
                       : RC == 8 ? (const void *)fps_coop_kernel<8, true> : (const void *)fps_coop_kernel<16, true>;
        int occ = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, FPS_THREADS, 0) != hipSuccess || occ < 1) occ = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 0;
        cap[slot] = std::max(1, occ * cus);
    }
    return cap[slot];
}

}  // namespace

extern "C" size_t farthest_point_sampling_hip_workspace_bytes(int b, int n_total) {
    (void)n_total;
    return sizeof(unsigned long long) * 2 * 4 * 256 * (size_t)(b > 0 ? b : 1) + sizeof(int) * (16 + COOP_CTRL_INTS) + 256;  // granule slots (key, x, y, z), error flag, team control block
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
    // cooperative variant: every cloud split over W <= 32 workgroups, all points in registers (2 .. 16 per lane): the fewest
    // points per lane that W allows (with the exchange through one XCD's L2 the per-lane update is what is left to shrink:
    // 120 k points as 30 x 4 per lane 1.42 us per sample, as 15 x 8 1.80, as 59 x 2 with two loads per poll 1.53)
    const int Wcap = std::min(32, 256 / b);  // 2 W granule pairs are polled by the 64 lanes of one wavefront
    int RC = 2;
    while (RC < 16 && (n_max + RC * FPS_THREADS - 1) / (RC * FPS_THREADS) > Wcap) RC *= 2;
    const int W = (n_max + RC * FPS_THREADS - 1) / (RC * FPS_THREADS);
    const bool coop = W >= 2 && W <= Wcap && n_max < (1 << 21) && workspace &&
                      workspace_bytes >= farthest_point_sampling_hip_workspace_bytes(b, n_total);
    if (coop) {
        unsigned long long *slots = (unsigned long long *)workspace;
        // one XCD per cloud (see the kernel): the launch carries enough workgroups that W - 1 of them may be left over on every
        // XCD, and all of it has to be resident at once; AO_AMD_FPS_LOCAL=0: the device-scope exchange (the tests' A/B switch)
        const char *le = getenv("AO_AMD_FPS_LOCAL");
        const int local_grid = (b - 1) * W + 8 * (W - 1) + 1;
        const bool local = !(le && le[0] == '0') && local_grid <= coop_capacity(RC) && (b + 7) / 8 + 1 <= COOP_TEAMS;
        int *err = (int *)((char *)workspace + sizeof(unsigned long long) * 2 * 4 * 256 * (size_t)b);
        int *ctrl = err + 16;
        (void)hipMemsetAsync(workspace, 0, sizeof(unsigned long long) * 2 * 4 * 256 * (size_t)b + sizeof(int) * (16 + COOP_CTRL_INTS), st);
#define FPS_COOP(RCV)                                                                                                            \
    do {                                                                                                                         \
        if (local)                                                                                                               \
            hipLaunchKernelGGL((fps_coop_kernel<RCV, true>), dim3(local_grid), dim3(FPS_THREADS), 0, st, xyz, offset, new_offset,  \
                               tmp, idx, B, logB, W, slots, err, b, ctrl);                                                       \
        else                                                                                                                     \
            hipLaunchKernelGGL((fps_coop_kernel<RCV, false>), dim3(b * W), dim3(FPS_THREADS), 0, st, xyz, offset, new_offset, tmp, \
                               idx, B, logB, W, slots, err, b, ctrl);                                                            \
    } while (0)
        if (RC == 2) FPS_COOP(2);
        else if (RC == 4) FPS_COOP(4);
        else if (RC == 16) FPS_COOP(16);
        else FPS_COOP(8);
#undef FPS_COOP
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    hipLaunchKernelGGL(fps_kernel, dim3(b), dim3(FPS_THREADS), 0, st, xyz, offset, new_offset, tmp, idx, B, logB);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
