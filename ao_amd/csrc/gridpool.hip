// ao_amd/csrc/gridpool.hip -- the coordinates-only half of GridPool.forward on the device
// (pointcept/models/point_transformer_v2/point_transformer_v2m2_base.py:246-268).
//
// The reference chains a python-loop offset2batch (host sync per cloud), torch_scatter.segment_csr(min),
// torch_geometric.voxel_grid, torch.unique (sync), torch.sort and two more segment_csr calls.  Here:
//   key[i]   = batch-major voxel id of point i      (torch_cluster.grid_cluster formula, same float ops)
//   sort     stable LSD radix sort of (key, i) over exactly the key's significant bits (rocPRIM via hipCUB --
//            a library sort, like the BLAS GEMMs; everything around it is hand-written)
//   heads    key[i] != key[i-1]  -> inclusive scan -> cluster rank; N' = number of clusters
//   outputs  cluster (N) fine->coarse map, order (N), idx_ptr (N'+1), pooled coordinates (N',3) as the mean of
//            the members in ascending point order (bit-identical to a sequential segment mean), new_offset (B)
// One 4-byte read-back (N') per pooling instead of ~8 host syncs.  Pooled features (segment max) are pool.hip.
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "common.h"

namespace {

constexpr int TPB = 256;
inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

struct PoolDims {
    long long nx, ny, nz;  // voxels per axis over the whole batch
    long long total;       // nx * ny * nz * clouds: must stay below 2^KEY_BITS
};
constexpr int KEY_BITS = 48;  // radix-sorted key width (6 passes); 2^48 voxels = (65536 per axis)^3

// per-axis max over clouds of (hi - lo), then trunc(/size) + 1   (grid_cluster: (end - start) / size + 1)
__device__ __forceinline__ PoolDims pool_dims_of(int b, const float *__restrict__ lo, const float *__restrict__ hi, float size) {
    float e[3] = {0.f, 0.f, 0.f};
    for (int s = 0; s < b; ++s)
        for (int a = 0; a < 3; ++a) e[a] = fmaxf(e[a], hi[3 * s + a] - lo[3 * s + a]);
    PoolDims d;
    d.nx = (long long)(e[0] / size) + 1;
    d.ny = (long long)(e[1] / size) + 1;
    d.nz = (long long)(e[2] / size) + 1;
    d.total = d.nx * d.ny * d.nz * (long long)b;
    return d;
}
__global__ void pool_dims_kernel(int b, const float *__restrict__ lo, const float *__restrict__ hi, float size,
                                 PoolDims *dims) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    *dims = pool_dims_of(b, lo, hi, size);
}

__global__ __launch_bounds__(TPB) void pool_keys_kernel(int n, int b, const float *__restrict__ coord,
                                                        const int *__restrict__ offset, const float *__restrict__ lo,
                                                        float size, const PoolDims *__restrict__ dims,
                                                        unsigned long long *__restrict__ keys, int *__restrict__ vals) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int s = seg_of(i, offset, b);
    const long long cx = (long long)((coord[3 * (size_t)i] - lo[3 * s]) / size);
    const long long cy = (long long)((coord[3 * (size_t)i + 1] - lo[3 * s + 1]) / size);
    const long long cz = (long long)((coord[3 * (size_t)i + 2] - lo[3 * s + 2]) / size);
    const long long nx = dims->nx, ny = dims->ny, nz = dims->nz;
    keys[i] = (unsigned long long)(cx + cy * nx + cz * nx * ny + (long long)s * nx * ny * nz);
    vals[i] = i;
}

__global__ __launch_bounds__(TPB) void pool_heads_kernel(int n, const unsigned long long *__restrict__ keys,
                                                         int *__restrict__ flags) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// rank[i] = inclusive scan of flags (1-based cluster number of sorted position i)
__global__ __launch_bounds__(TPB) void pool_scatter_kernel(int n, const int *__restrict__ rank,
                                                           const int *__restrict__ flags, const int *__restrict__ order,
                                                           long long *__restrict__ cluster, int *__restrict__ idx_ptr,
                                                           int *__restrict__ n_out, const PoolDims *__restrict__ dims) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int r = rank[i] - 1;
    cluster[order[i]] = r;
    if (flags[i]) idx_ptr[r] = i;
    if (i == n - 1) {
        idx_ptr[r + 1] = n;
        *n_out = (dims->total > 0 && dims->total < (1ll << KEY_BITS)) ? r + 1 : -1;  // -1: voxel ids exceed the sort key
    }
}

// pooled coordinate = sequential fp32 sum over the members in sorted (= ascending point) order / count;
// new_offset[s] = number of clusters whose cloud index is <= s
// sort_members (dense path): the member slots came from atomics -- the cluster's thread first puts its handful of members
// (one voxel) into ascending point order, in place, so that `order` is the stable sort by key and the sum below runs in it
// `over` (dense path): set by the scan when some voxel holds more than DENSE_MAX_MEMBERS points -- the one-thread insertion
// sort below is quadratic in a voxel's members (a coarse grid, a duplicate-heavy cloud, a scene inside ONE voxel: 10^5 members
// = 10^10 serial global read-modify-writes), so such a call is handed back (*n_out = -2) and repeated on the radix-sort path,
// whose cost does not depend on the occupancy
__global__ __launch_bounds__(TPB) void pool_mean_kernel(int *n_out, int b,
                                                        const float *__restrict__ coord, const int *__restrict__ offset,
                                                        int *order, const int *__restrict__ idx_ptr,
                                                        float *__restrict__ new_coord, int *__restrict__ new_offset, int sort_members,
                                                        const int *__restrict__ over) {
    if (over && *over) {
        // (every thread leaves before anyone reads *n_out below: the store cannot race with a read of this launch)
        if (blockIdx.x == 0 && threadIdx.x == 0 && *n_out >= 0) *n_out = -2;
        return;
    }
    const int m = *n_out;  // negative on key overflow: nothing to do
    for (int j = blockIdx.x * TPB + threadIdx.x; j < m; j += gridDim.x * TPB) {
        const int p0 = idx_ptr[j], p1 = idx_ptr[j + 1];
        if (sort_members) {
            for (int p = p0 + 1; p < p1; ++p) {
                const int v = order[p];
                int q = p - 1;
                while (q >= p0 && order[q] > v) { order[q + 1] = order[q]; --q; }
                order[q + 1] = v;
            }
        }
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int p = p0; p < p1; ++p) {
            const size_t r = (size_t)order[p];
            sx += coord[3 * r]; sy += coord[3 * r + 1]; sz += coord[3 * r + 2];
        }
        const float cnt = (float)(p1 - p0);
        new_coord[3 * (size_t)j] = sx / cnt;
        new_coord[3 * (size_t)j + 1] = sy / cnt;
        new_coord[3 * (size_t)j + 2] = sz / cnt;
        const int s0 = seg_of(order[p0], offset, b);
        // (the next cluster's first slot, possibly while its own thread is still sorting it: every value ever stored in that range
        // is one of its members, and any member lies in the cluster's cloud)
        const int s1 = (j + 1 < m) ? seg_of(order[idx_ptr[j + 1]], offset, b) : b;
        for (int s = s0; s < s1; ++s) new_offset[s] = j + 1;  // clouds s0 .. s1-1 end after cluster j
        if (j == 0)
            for (int s = 0; s < s0; ++s) new_offset[s] = 0;    // empty leading clouds
    }
}

// ------------------------------------------------------------------ dense path --
// When the whole voxel grid (nx * ny * nz * clouds cells) is small enough to tabulate -- DENSE_CAP cells: every S3DIS /
// ScanNet pooling level (0.06 m voxels over a 10 x 10 x 3 m scan are 1.4 M cells) -- no sort is needed at all:
//   count[key]++ (the returned old value is the point's slot inside its voxel), an exclusive scan of the table gives every
//   voxel its cluster number (occupied voxels before it, in key order = the order a sort + unique produces) and the start of
//   its member list, the points are dropped into their slots, and each cluster orders its handful of members by point
//   index (the slots came from atomics) before it takes their mean -- so `order` is exactly the stable sort by key.
// Six launches (the member sort rides in the mean kernel) instead of the ~16 of a 6-pass radix sort + scan, and 40 us instead of 250 at 120 k points.  The grid size
// is only known on the device: a grid beyond DENSE_CAP sets *n_out = -2 and the caller repeats the call on the sort path.
constexpr long long DENSE_CAP = 1ll << 23;
constexpr int DENSE_MAX_MEMBERS = 64;  // members of one voxel the dense path sorts with one thread (PT-v2 levels: 5-15)
constexpr int DSCAN_THREADS = 256, DSCAN_ITEMS = 8, DSCAN_TILE = DSCAN_THREADS * DSCAN_ITEMS;

// (the grid's dimensions are derived here by every workgroup -- a loop over the b clouds' extents -- and stored by the first:
// what was a one-thread launch of its own in front of this one)
__global__ __launch_bounds__(TPB) void dense_zero_kernel(int b, const float *__restrict__ lo, const float *__restrict__ hi, float size,
                                                         PoolDims *__restrict__ dims, int *__restrict__ count, int *n_out,
                                                         long long cells_cap, int *__restrict__ over) {
    const PoolDims d = pool_dims_of(b, lo, hi, size);
    if (blockIdx.x == 0 && threadIdx.x == 0) { *dims = d; *over = 0; }
    const long long total = d.total;
    if (total <= 0 || total > cells_cap) {  // (beyond the table carved for this n: the sort path)
        if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = (total > 0 && total < (1ll << KEY_BITS)) ? -2 : -1;
        return;
    }
    const long long pad = (total + DSCAN_TILE - 1) / DSCAN_TILE * DSCAN_TILE;  // (the scan reads whole tiles)
    const int4 z = make_int4(0, 0, 0, 0);
    for (long long i = ((long long)blockIdx.x * TPB + threadIdx.x) * 4; i < pad; i += (long long)gridDim.x * TPB * 4) *(int4 *)(count + i) = z;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = 0;
}

__global__ __launch_bounds__(TPB) void dense_keys_kernel(int n, int b, const float *__restrict__ coord, const int *__restrict__ offset,
                                                         const float *__restrict__ lo, float size, const PoolDims *__restrict__ dims,
                                                         const int *__restrict__ n_out, int *__restrict__ count,
                                                         int *__restrict__ key32, int *__restrict__ slot) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n || *n_out < 0) return;
    const int s = seg_of(i, offset, b);
    const long long cx = (long long)((coord[3 * (size_t)i] - lo[3 * s]) / size);
    const long long cy = (long long)((coord[3 * (size_t)i + 1] - lo[3 * s + 1]) / size);
    const long long cz = (long long)((coord[3 * (size_t)i + 2] - lo[3 * s + 2]) / size);
    const long long nx = dims->nx, ny = dims->ny, nz = dims->nz;
    const int key = (int)(cx + cy * nx + cz * nx * ny + (long long)s * nx * ny * nz);
    key32[i] = key;
    slot[i] = atomicAdd(&count[key], 1);
}

// two exclusive scans over the cell table at once: members before a cell (x) and occupied cells before it (y)
__global__ __launch_bounds__(DSCAN_THREADS) void dense_scan_reduce_kernel(const PoolDims *__restrict__ dims, const int *__restrict__ n_out,
                                                                          const int *__restrict__ count, int2 *__restrict__ tile_sums) {
    if (*n_out < 0 || (long long)blockIdx.x * DSCAN_TILE >= dims->total) return;
    __shared__ int2 s_w[DSCAN_THREADS / WAVE];
    const int4 *p = (const int4 *)(count + (size_t)blockIdx.x * DSCAN_TILE) + threadIdx.x * 2;
    const int4 a = p[0], c = p[1];
    int v = a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w;
    int o = (a.x > 0) + (a.y > 0) + (a.z > 0) + (a.w > 0) + (c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { v += __shfl_xor(v, d, WAVE); o += __shfl_xor(o, d, WAVE); }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = make_int2(v, o);
    __syncthreads();
    if (threadIdx.x == 0) {
        int2 t = make_int2(0, 0);
        for (int i = 0; i < DSCAN_THREADS / WAVE; ++i) { t.x += s_w[i].x; t.y += s_w[i].y; }
        tile_sums[blockIdx.x] = t;
    }
}

// base[cell] = (members before the cell, occupied cells before it); the last tile also publishes the cluster count
__global__ __launch_bounds__(DSCAN_THREADS) void dense_scan_apply_kernel(const PoolDims *__restrict__ dims, int *n_out, int n,
                                                                         const int *__restrict__ count, const int2 *__restrict__ tile_sums,
                                                                         int2 *__restrict__ base, int *__restrict__ idx_ptr,
                                                                         int *__restrict__ over) {
    const long long total = dims->total;
    if (*n_out < 0 || (long long)blockIdx.x * DSCAN_TILE >= total) return;
    __shared__ int2 s_w[DSCAN_THREADS / WAVE];
    __shared__ int2 s_base;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int2 part = make_int2(0, 0);
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += DSCAN_THREADS) { const int2 t = tile_sums[i]; part.x += t.x; part.y += t.y; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { part.x += __shfl_xor(part.x, d, WAVE); part.y += __shfl_xor(part.y, d, WAVE); }
    if (lane == 0) s_w[wid] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        int2 t = make_int2(0, 0);
        for (int i = 0; i < DSCAN_THREADS / WAVE; ++i) { t.x += s_w[i].x; t.y += s_w[i].y; }
        s_base = t;
    }
    __syncthreads();
    const int2 tb = s_base;
    __syncthreads();
    const int4 *p = (const int4 *)(count + (size_t)blockIdx.x * DSCAN_TILE) + threadIdx.x * 2;
    const int4 a = p[0], c = p[1];
    const int v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
    int ex[8], eo[8], tot = 0, occ = 0, big = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ex[i] = tot; eo[i] = occ; tot += v[i]; occ += v[i] > 0; big |= v[i] > DENSE_MAX_MEMBERS; }
    if (big) *over = 1;  // (benign race: every writer stores the same value)
    int inc = tot, inco = occ;  // inclusive wave scans of the per-thread totals
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const int y = __shfl_up(inc, d, WAVE), yo = __shfl_up(inco, d, WAVE);
        if (lane >= d) { inc += y; inco += yo; }
    }
    if (lane == 63) s_w[wid] = make_int2(inc, inco);
    __syncthreads();
    int wx = 0, wo = 0;
    for (int i = 0; i < wid; ++i) { wx += s_w[i].x; wo += s_w[i].y; }
    const int bx = tb.x + wx + inc - tot, bo = tb.y + wo + inco - occ;
    int2 *q = base + (size_t)blockIdx.x * DSCAN_TILE + (size_t)threadIdx.x * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = make_int2(bx + ex[i], bo + eo[i]);
    const long long last_tile = (total - 1) / DSCAN_TILE;
    if ((long long)blockIdx.x == last_tile && threadIdx.x == DSCAN_THREADS - 1) {
        const int m = bo + occ;  // (cells past `total` inside the last tile are zero)
        *n_out = m;
        idx_ptr[m] = n;
    }
}

__global__ __launch_bounds__(TPB) void dense_scatter_kernel(int n, const int *__restrict__ n_out, const int *__restrict__ key32,
                                                            const int *__restrict__ slot, const int2 *__restrict__ base,
                                                            long long *__restrict__ cluster, int *__restrict__ order,
                                                            int *__restrict__ idx_ptr) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n || *n_out < 0) return;
    const int2 bb = base[key32[i]];
    cluster[i] = bb.y;
    order[bb.x + slot[i]] = i;
    if (slot[i] == 0) idx_ptr[bb.y] = bb.x;
}

struct Ws {
    float *lo, *hi;
    PoolDims *dims;
    unsigned long long *keys_in, *keys_out;
    int *vals_in, *flags, *rank;
    void *mm, *cub;
    size_t mm_bytes, cub_bytes, bytes;
    // dense path
    int *d_count, *d_key, *d_slot, *d_over;
    int2 *d_tiles, *d_base;
    long long d_cells;  // cells the dense tables were carved for
};

}  // namespace

extern "C" size_t segment_minmax_hip_workspace_bytes(int b);
extern "C" int segment_minmax_hip_launcher(int b, const float *xyz, const int *offset, float *lo, float *hi,
                                           void *workspace, size_t workspace_bytes, void *stream);

static Ws carve(void *base, int n, int b) {
    Ws w;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += al(bytes); return r; };
    w.lo = (float *)take(sizeof(float) * 3 * b);
    w.hi = (float *)take(sizeof(float) * 3 * b);
    w.dims = (PoolDims *)take(sizeof(PoolDims));
    w.keys_in = (unsigned long long *)take(sizeof(unsigned long long) * n);
    w.keys_out = (unsigned long long *)take(sizeof(unsigned long long) * n);
    w.vals_in = (int *)take(sizeof(int) * n);
    w.flags = (int *)take(sizeof(int) * n);
    w.rank = (int *)take(sizeof(int) * n);
    w.mm_bytes = segment_minmax_hip_workspace_bytes(b);
    w.mm = take(w.mm_bytes);
    size_t s1 = 0, s2 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, s1, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                             (const int *)nullptr, (int *)nullptr, n, 0, 64, (hipStream_t)0);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, s2, (const int *)nullptr, (int *)nullptr, n, (hipStream_t)0);
    w.cub_bytes = std::max(s1, s2) + 256;
    w.cub = take(w.cub_bytes);
    // dense path: the grid size is data dependent, so the table is sized from n -- 64 cells per point (PT-v2 levels have 3-12:
    // a 0.06 m grid over a 120 k-point scan is 1.4 M cells), at most DENSE_CAP -- and a grid beyond it takes the sort path.
    // (Sized for the cap whatever n, the tables were 100 MB of every stream's retained workspace, also for a 2 k-point call.)
    w.d_cells = std::min<long long>(DENSE_CAP, std::max<long long>(64ll * n, 1 << 16));
    const size_t cells = (size_t)w.d_cells + DSCAN_TILE;
    w.d_count = (int *)take(sizeof(int) * cells);
    w.d_base = (int2 *)take(sizeof(int2) * cells);
    w.d_tiles = (int2 *)take(sizeof(int2) * (cells / DSCAN_TILE + 1));
    w.d_key = (int *)take(sizeof(int) * n);
    w.d_slot = (int *)take(sizeof(int) * n);
    w.d_over = (int *)take(sizeof(int) * 4);
    w.bytes = off;
    return w;
}

extern "C" size_t grid_pool_hip_workspace_bytes(int n, int b) {
    if (n < 1 || b < 1) return 0;
    return carve(nullptr, n, b).bytes + 1024;
}

// cluster (n) int64, order (n) int32, idx_ptr (n+1) int32 [first n_out+1 entries valid], new_coord (n,3) and
// new_offset (b) [first n_out rows valid], n_out (1) int32 -- all device pointers; the caller reads n_out back.
// sort_path == 0: the dense path; *n_out == -2 afterwards means the voxel grid is too large to tabulate -- call again with
// sort_path != 0 (radix sort of the keys).  *n_out == -1: voxel ids exceed the 48-bit sort key.
extern "C" int grid_pool_hip_launcher(int n, int b, const float *coord, const int *offset, float grid_size,
                                      long long *cluster, int *order, int *idx_ptr, float *new_coord,
                                      int *new_offset, int *n_out, int sort_path, void *workspace, size_t workspace_bytes,
                                      void *stream) {
    if (n < 1 || b < 1 || !(grid_size > 0.f)) return PTV2_ERR_ARG;
    Ws w = carve(workspace, n, b);
    if (!workspace || workspace_bytes < w.bytes) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int rc = segment_minmax_hip_launcher(b, coord, offset, w.lo, w.hi, w.mm, w.mm_bytes, stream);
    if (rc != PTV2_OK) return rc;
    const int nb = divup(n, TPB);
    if (sort_path)
        hipLaunchKernelGGL(pool_dims_kernel, dim3(1), dim3(64), 0, st, b, (const float *)w.lo, (const float *)w.hi, grid_size,
                           w.dims);
    if (!sort_path) {
        const int ztiles = (int)((w.d_cells + DSCAN_TILE) / DSCAN_TILE);
        hipLaunchKernelGGL(dense_zero_kernel, dim3(1024), dim3(TPB), 0, st, b, (const float *)w.lo, (const float *)w.hi, grid_size,
                           w.dims, w.d_count, n_out, w.d_cells, w.d_over);
        hipLaunchKernelGGL(dense_keys_kernel, dim3(nb), dim3(TPB), 0, st, n, b, coord, offset, (const float *)w.lo, grid_size,
                           (const PoolDims *)w.dims, (const int *)n_out, w.d_count, w.d_key, w.d_slot);
        hipLaunchKernelGGL(dense_scan_reduce_kernel, dim3(ztiles), dim3(DSCAN_THREADS), 0, st, (const PoolDims *)w.dims,
                           (const int *)n_out, (const int *)w.d_count, w.d_tiles);
        hipLaunchKernelGGL(dense_scan_apply_kernel, dim3(ztiles), dim3(DSCAN_THREADS), 0, st, (const PoolDims *)w.dims, n_out, n,
                           (const int *)w.d_count, (const int2 *)w.d_tiles, w.d_base, idx_ptr, w.d_over);
        hipLaunchKernelGGL(dense_scatter_kernel, dim3(nb), dim3(TPB), 0, st, n, (const int *)n_out, (const int *)w.d_key,
                           (const int *)w.d_slot, (const int2 *)w.d_base, cluster, order, idx_ptr);
        hipLaunchKernelGGL(pool_mean_kernel, dim3(std::min(nb, 2048)), dim3(TPB), 0, st, n_out, b, coord, offset,
                           order, (const int *)idx_ptr, new_coord, new_offset, 1, (const int *)w.d_over);
        PTV2_CHECK_LAUNCH();
        return PTV2_OK;
    }
    hipLaunchKernelGGL(pool_keys_kernel, dim3(nb), dim3(TPB), 0, st, n, b, coord, offset, (const float *)w.lo, grid_size,
                       (const PoolDims *)w.dims, w.keys_in, w.vals_in);
    size_t cb = w.cub_bytes;
    // KEY_BITS key bits: the voxel counts live on the device, so the range is fixed (and checked) instead of trimmed
    if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, (const unsigned long long *)w.keys_in, w.keys_out,
                                           (const int *)w.vals_in, order, n, 0, KEY_BITS, st) != hipSuccess)
        return PTV2_ERR_LAUNCH;
    hipLaunchKernelGGL(pool_heads_kernel, dim3(nb), dim3(TPB), 0, st, n, (const unsigned long long *)w.keys_out, w.flags);
    cb = w.cub_bytes;
    if (hipcub::DeviceScan::InclusiveSum(w.cub, cb, (const int *)w.flags, w.rank, n, st) != hipSuccess) return PTV2_ERR_LAUNCH;
    hipLaunchKernelGGL(pool_scatter_kernel, dim3(nb), dim3(TPB), 0, st, n, (const int *)w.rank, (const int *)w.flags,
                       (const int *)order, cluster, idx_ptr, n_out, (const PoolDims *)w.dims);
    hipLaunchKernelGGL(pool_mean_kernel, dim3(std::min(nb, 2048)), dim3(TPB), 0, st, n_out, b, coord, offset,
                       order, (const int *)idx_ptr, new_coord, new_offset, 0, (const int *)nullptr);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}


