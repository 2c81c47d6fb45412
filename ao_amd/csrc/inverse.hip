// ao_amd/csrc/inverse.hip -- inverse neighbour tables (CSR) of a whole scene's tables in four launches.
//
// For a neighbour table idx (n, k) with entries in [-1, n): for every point j the slots r = i*k + s with idx[r] == j, in
// ascending r: inv_rows[inv_ptr[j] .. inv_ptr[j+1]); the -1 placeholders are inv_rows[0 .. inv_ptr[0]).  The backward kernels
// of the attention (gva_bwd.hip, gva_aggregate.hip) and of the interpolation (gather_ops.hip) walk these lists to turn the
// reference's atomicAdd scatters (pointops/src/attention/attention_cuda_kernel.cu:37-66, interpolation_cuda_kernel.cu:33-45)
// into gathers that sum in a fixed order.
//
// Rounds 1-3 built each table with a library radix sort of (idx + 1, r) plus a binary search per point: 13 launches and
// ~130 us per table, seven tables per scene -- 91 of the geometry's 202 launches and half of its time.  The keys are small
// integers (key = idx + 1 in [0, n]) and a bucket has k members on average, so this is a two-level counting sort, for all
// tables of a scene side by side in one index space (jobs), with no atomics on global memory (a first version counted and
// scattered with one device-scope atomic per slot: 270 us for the 4.1 M slots of the bench scene, and it cost the training
// step running beside it 0.21 ms -- device-scope atomics execute at the memory side):
//   split   every workgroup counts the slots of its 4 096-slot chunk per PARTITION (<= 512 runs of `width` consecutive keys per
//           table; LDS atomics) and writes the counts partition-major
//   scan    one chained exclusive scan over (job, partition, chunk): every workgroup publishes its total, workgroup b adds the
//           totals of 0 .. b-1 (workgroups start in index order, so the ones waited for are running or done)
//   spread  the same chunks again: every slot takes the next place of its partition's run for this chunk (LDS cursor) and
//           stores (slot, key) there -- partitions complete and contiguous, order inside arbitrary
//   build   one workgroup per partition: counts its `width` buckets in LDS, scans them (-> inv_ptr), places the members
//           (LDS, or global scratch when a partition is larger than the LDS staging), then sixteen lanes per bucket rank its
//           members (all distinct) and write them ascending (-> inv_rows)
// Results are bit-identical to the sort's (tests/test_gpu_gva_stages.py against the host statement).
#include <algorithm>

#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int INV_MAX = PTV2_INVERSE_MAX_JOBS;
constexpr int CHUNK = 4096;                 // slots per workgroup of split / spread
constexpr int MAX_PARTS = 512;              // partitions per job
constexpr int MIN_WIDTH = 256, MAX_WIDTH = 4096;  // keys per partition (a power of two)
constexpr int STAGE = 12288;                // members a partition stages in LDS (48 KB); larger ones go through global scratch
constexpr int SCAN_TPB = 1024, SCAN_ITEMS = 4, SCAN_TILE = SCAN_TPB * SCAN_ITEMS;
constexpr int GROUP = 16, GROUPS = TPB / GROUP;
inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

struct InvJobs {
    int count;
    int n[INV_MAX];             // points of job t (keys 0 .. n: key 0 = the -1 placeholders)
    const int *idx[INV_MAX];
    int *inv_ptr[INV_MAX];
    int *inv_rows[INV_MAX];
    int rows[INV_MAX];          // n * k
    int shift[INV_MAX];         // partition of a key = key >> shift (width = 1 << shift)
    int parts[INV_MAX];         // partitions of job t
    int chunks[INV_MAX];        // chunks of job t
    int row0[INV_MAX + 1];      // first packed position of job t's lists (sum of the rows before it)
    int chunk0[INV_MAX + 1];    // first chunk (= workgroup of split / spread) of job t
    int part0[INV_MAX + 1];     // first partition (= workgroup of build) of job t
    int cell0[INV_MAX + 1];     // first (partition, chunk) counter of job t: cell = cell0 + partition * chunks + chunk
};

__device__ __forceinline__ int job_of(const int *starts, int count, int v) {
    int t = 0;
    while (t + 1 < count && v >= starts[t + 1]) ++t;
    return t;
}

// key of a slot: 0 for a placeholder (or anything outside the table), j + 1 for neighbour j
__device__ __forceinline__ int key_of(int v, int n) { return ((unsigned)v < (unsigned)n) ? v + 1 : 0; }

// SPREAD = false: cells[job][partition][chunk] = slots of the chunk whose key falls into the partition
// SPREAD = true:  cells hold the exclusive scan of that; every slot stores (slot, key) at its partition's next place
template <bool SPREAD>
__global__ __launch_bounds__(TPB) void inv_split_kernel(InvJobs J, int *__restrict__ cells, int2 *__restrict__ pairs,
                                                        unsigned long long *__restrict__ handoff, int handoff_words,
                                                        int cells_used, int cells_padded) {
    __shared__ int s_part[MAX_PARTS];
    const int t = job_of(J.chunk0, J.count, blockIdx.x);
    const int chunk = blockIdx.x - J.chunk0[t], parts = J.parts[t], chunks = J.chunks[t], shift = J.shift[t], n = J.n[t];
    int *cell = cells + J.cell0[t] + chunk;  // + partition * chunks
    if (!SPREAD && blockIdx.x == 0) {  // (the scan's hand-off words and the padding behind the last cell)
        for (int i = threadIdx.x; i < handoff_words; i += TPB) handoff[i] = 0ull;
        for (int i = cells_used + threadIdx.x; i < cells_padded; i += TPB) cells[i] = 0;
    }
    for (int p = threadIdx.x; p < parts; p += TPB) s_part[p] = SPREAD ? cell[(size_t)p * chunks] : 0;
    __syncthreads();
    const int *idx = J.idx[t];
    const int r0 = chunk * CHUNK, r1 = min(r0 + CHUNK, J.rows[t]);
    for (int r = r0 + threadIdx.x; r < r1; r += TPB) {
        const int key = key_of(idx[r], n);
        const int place = atomicAdd(&s_part[key >> shift], 1);
        if (SPREAD) pairs[place] = make_int2(r, key);
    }
    if (!SPREAD) {
        __syncthreads();
        for (int p = threadIdx.x; p < parts; p += TPB) cell[(size_t)p * chunks] = s_part[p];
    }
}

// exclusive prefix over all cells, in place.  Hand-off word of workgroup b: (1 << 32) | its total.
__global__ __launch_bounds__(SCAN_TPB) void inv_scan_kernel(int *__restrict__ cells, unsigned long long *__restrict__ handoff) {
    __shared__ int s_wave[SCAN_TPB / 64];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, b = blockIdx.x;
    int4 *at = (int4 *)(cells + (size_t)b * SCAN_TILE + tid * SCAN_ITEMS);
    const int4 v = *at;  // (the cell array is padded to a whole tile with zeros)
    const int tsum = v.x + v.y + v.z + v.w;
    int incl = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wid] = incl;
    __syncthreads();
    int wave_excl = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_TPB / 64; ++w) {
        const int s = s_wave[w];
        if (w < wid) wave_excl += s;
        total += s;
    }
    if (tid == 0)
        __hip_atomic_store(handoff + b, (1ull << 32) | (unsigned)total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int before = 0;
    for (int i = tid; i < b; i += SCAN_TPB) {
        unsigned long long w;
        do {
            w = __hip_atomic_load(handoff + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (!(w >> 32)) __builtin_amdgcn_s_sleep(1);
        } while (!(w >> 32));
        before += (int)(unsigned)w;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
    __syncthreads();  // (s_wave was read above by everyone)
    if (lane == 0) s_wave[wid] = before;
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        for (int w = 0; w < SCAN_TPB / 64; ++w) s += s_wave[w];
        s_base = s;
    }
    __syncthreads();
    const int e0 = s_base + wave_excl + incl - tsum;
    *at = make_int4(e0, e0 + v.x, e0 + v.x + v.y, e0 + v.x + v.y + v.z);
}

// members [m0, m1) of one bucket, arrival order in `src`; written ascending to dst[0 .. s)
template <typename Src>
__device__ __forceinline__ void order_bucket(Src src, int s, int *__restrict__ dst, int l) {
    if (s <= GROUP) {  // one member per lane, the others through the lanes of the group
        const int mine = l < s ? src(l) : 0x7fffffff;
        int rank = 0;
#pragma unroll
        for (int q = 0; q < GROUP; ++q) rank += __shfl(mine, q, GROUP) < mine;
        if (l < s) dst[rank] = mine;
        return;
    }
    for (int e = l; e < s; e += GROUP) {  // s * ceil(s / 16) compares per lane
        const int mine = src(e);
        int rank = 0;
        for (int q = 0; q < s; ++q) rank += src(q) < mine;
        dst[rank] = mine;
    }
}

// one workgroup per partition (job t, keys [p << shift, (p + 1) << shift)): pairs[P0, P1) are its members
__global__ __launch_bounds__(TPB) void inv_build_kernel(InvJobs J, const int *__restrict__ cells, const int2 *__restrict__ pairs,
                                                        int *__restrict__ spill) {
    extern __shared__ int lds[];
    const int t = job_of(J.part0, J.count, blockIdx.x);
    const int p = blockIdx.x - J.part0[t], shift = J.shift[t], width = 1 << shift, n = J.n[t], chunks = J.chunks[t];
    int *s_start = lds;               // [width + 1] exclusive prefix of the bucket sizes
    int *s_cur = lds + width + 1;     // [width]     counters, then cursors
    int *s_stage = s_cur + width;     // [STAGE]
    __shared__ int s_wave[TPB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int *cell = cells + J.cell0[t] + (size_t)p * chunks;
    const int P0 = cell[0];
    const int P1 = (p + 1 < J.parts[t]) ? cell[chunks] : J.row0[t] + J.rows[t];
    const int S = P1 - P0, key0 = p << shift;
    for (int i = tid; i < width; i += TPB) s_cur[i] = 0;
    __syncthreads();
    for (int i = tid; i < S; i += TPB) atomicAdd(&s_cur[pairs[P0 + i].y - key0], 1);
    __syncthreads();
    // exclusive scan of the `width` counters: every thread takes width / TPB consecutive ones
    const int per = width / TPB;  // 1 .. 16
    int c[MAX_WIDTH / TPB], tsum = 0;
#pragma unroll
    for (int q = 0; q < MAX_WIDTH / TPB; ++q)
        if (q < per) { c[q] = s_cur[tid * per + q]; tsum += c[q]; }
    int incl = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wid] = incl;
    __syncthreads();
    int run = incl - tsum;
    for (int w = 0; w < wid; ++w) run += s_wave[w];
    int *inv_ptr = J.inv_ptr[t];
    const int base = P0 - J.row0[t];  // this partition's first position in the job's inv_rows
#pragma unroll
    for (int q = 0; q < MAX_WIDTH / TPB; ++q)
        if (q < per) {
            const int kl = tid * per + q;
            s_start[kl] = run;
            s_cur[kl] = run;
            run += c[q];
            if (key0 + kl <= n) inv_ptr[key0 + kl] = base + run;  // end of bucket `key` = start of the list of point `key`
        }
    if (tid == TPB - 1) s_start[width] = run;
    __syncthreads();
    const bool staged = S <= STAGE;  // (uniform)
    int *place = staged ? s_stage : spill + P0;
    if (staged) {
        for (int i = tid; i < S; i += TPB) {
            const int2 m = pairs[P0 + i];
            s_stage[atomicAdd(&s_cur[m.y - key0], 1)] = m.x;
        }
    } else {
        for (int i = tid; i < S; i += TPB) {
            const int2 m = pairs[P0 + i];
            place[atomicAdd(&s_cur[m.y - key0], 1)] = m.x;
        }
        __threadfence_block();
    }
    __syncthreads();
    int *inv_rows = J.inv_rows[t] + base;
    const int l = tid & (GROUP - 1);
    for (int kl = tid >> 4; kl < width; kl += GROUPS) {
        const int m0 = s_start[kl], s = s_start[kl + 1] - m0;
        if (s == 0) continue;
        if (staged) {
            const int *src = s_stage + m0;
            order_bucket([src](int e) { return src[e]; }, s, inv_rows + m0, l);
        } else {
            const int *src = place + m0;
            order_bucket([src](int e) { return __builtin_nontemporal_load(src + e); }, s, inv_rows + m0, l);
        }
    }
}

struct Plan {
    InvJobs J;
    int total_chunks, total_parts, cells_used, cells_padded, scan_wgs, max_width;
    long long total_rows;
    size_t cell_bytes, handoff_bytes, pair_bytes, spill_bytes;
};

int make_plan(int count, const ptv2_inverse_job *jobs, Plan &P) {
    if (count < 1 || count > INV_MAX || !jobs) return PTV2_ERR_ARG;
    InvJobs &J = P.J;
    J.count = count;
    long long rows = 0, chunks = 0, parts = 0, cells = 0;
    P.max_width = MIN_WIDTH;
    for (int t = 0; t < count; ++t) {
        const ptv2_inverse_job &j = jobs[t];
        if (j.n < 1 || j.k < 1) return PTV2_ERR_ARG;
        const long long r = (long long)j.n * j.k;
        int shift = 8;  // MIN_WIDTH
        while ((((long long)j.n + 1 + (1 << shift) - 1) >> shift) > MAX_PARTS) ++shift;
        if ((1 << shift) > MAX_WIDTH) return PTV2_ERR_ARG;  // more than 2 M points in one table
        P.max_width = std::max(P.max_width, 1 << shift);
        J.n[t] = j.n; J.idx[t] = j.idx; J.inv_ptr[t] = j.inv_ptr; J.inv_rows[t] = j.inv_rows;
        J.rows[t] = (int)r;
        J.shift[t] = shift;
        J.parts[t] = (int)(((long long)j.n + 1 + (1 << shift) - 1) >> shift);
        J.chunks[t] = (int)((r + CHUNK - 1) / CHUNK);
        J.row0[t] = (int)rows; J.chunk0[t] = (int)chunks; J.part0[t] = (int)parts; J.cell0[t] = (int)cells;
        rows += r;
        chunks += J.chunks[t];
        parts += J.parts[t];
        cells += (long long)J.parts[t] * J.chunks[t];
        if (rows >= (1ll << 31) || cells >= (1ll << 30)) return PTV2_ERR_ARG;
    }
    for (int t = count; t <= INV_MAX; ++t) {
        J.row0[t] = (int)rows; J.chunk0[t] = (int)chunks; J.part0[t] = (int)parts; J.cell0[t] = (int)cells;
        if (t < INV_MAX) {
            J.n[t] = 0; J.rows[t] = 0; J.shift[t] = 8; J.parts[t] = 0; J.chunks[t] = 0;
            J.idx[t] = nullptr; J.inv_ptr[t] = nullptr; J.inv_rows[t] = nullptr;
        }
    }
    P.total_rows = rows;
    P.total_chunks = (int)chunks;
    P.total_parts = (int)parts;
    P.cells_used = (int)cells;
    P.scan_wgs = (int)((cells + SCAN_TILE - 1) / SCAN_TILE);
    P.cells_padded = P.scan_wgs * SCAN_TILE;
    P.cell_bytes = al(sizeof(int) * (size_t)P.cells_padded);
    P.handoff_bytes = al(sizeof(unsigned long long) * (size_t)P.scan_wgs);
    P.pair_bytes = al(sizeof(int2) * (size_t)rows);
    P.spill_bytes = al(sizeof(int) * (size_t)rows);
    return PTV2_OK;
}

size_t build_lds_bytes(int width) { return sizeof(int) * ((size_t)2 * width + 1 + STAGE); }

}  // namespace

extern "C" size_t inverse_tables_hip_workspace_bytes(int count, const ptv2_inverse_job *jobs) {
    Plan P;
    if (make_plan(count, jobs, P) != PTV2_OK) return 0;
    return P.cell_bytes + P.handoff_bytes + P.pair_bytes + P.spill_bytes + 1024;
}

extern "C" int inverse_tables_hip_launcher(int count, const ptv2_inverse_job *jobs, void *workspace, size_t workspace_bytes,
                                           void *stream) {
    Plan P;
    const int rc = make_plan(count, jobs, P);
    if (rc != PTV2_OK) return rc;
    for (int t = 0; t < count; ++t)
        if (!jobs[t].idx || !jobs[t].inv_ptr || !jobs[t].inv_rows) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < P.cell_bytes + P.handoff_bytes + P.pair_bytes + P.spill_bytes) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char *p = (char *)workspace;
    int *cells = (int *)p;                                     p += P.cell_bytes;
    unsigned long long *handoff = (unsigned long long *)p;     p += P.handoff_bytes;
    int2 *pairs = (int2 *)p;                                   p += P.pair_bytes;
    int *spill = (int *)p;
    const size_t lds = build_lds_bytes(P.max_width);
    static bool raised = [] {
        return hipFuncSetAttribute((const void *)inv_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)build_lds_bytes(MAX_WIDTH)) == hipSuccess;
    }();
    if (!raised) return PTV2_ERR_LAUNCH;
    const dim3 chunks((unsigned)P.total_chunks);
    hipLaunchKernelGGL(inv_split_kernel<false>, chunks, dim3(TPB), 0, st, P.J, cells, pairs, handoff, P.scan_wgs, P.cells_used,
                       P.cells_padded);
    hipLaunchKernelGGL(inv_scan_kernel, dim3((unsigned)P.scan_wgs), dim3(SCAN_TPB), 0, st, cells, handoff);
    hipLaunchKernelGGL(inv_split_kernel<true>, chunks, dim3(TPB), 0, st, P.J, cells, pairs, handoff, P.scan_wgs, P.cells_used,
                       P.cells_padded);
    hipLaunchKernelGGL(inv_build_kernel, dim3((unsigned)P.total_parts), dim3(TPB), lds, st, P.J, (const int *)cells,
                       (const int2 *)pairs, spill);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" size_t inverse_table_hip_workspace_bytes(int n, int k) {
    if (n < 1 || k < 1) return 0;
    const ptv2_inverse_job job = {n, k, nullptr, nullptr, nullptr};
    return inverse_tables_hip_workspace_bytes(1, &job);
}

extern "C" int inverse_table_hip_launcher(int n, int k, const int *idx, int *inv_ptr, int *inv_rows, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    const ptv2_inverse_job job = {n, k, idx, inv_ptr, inv_rows};
    return inverse_tables_hip_launcher(1, &job, workspace, workspace_bytes, stream);
}
