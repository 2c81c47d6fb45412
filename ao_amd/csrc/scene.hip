// ao_amd/csrc/scene.hip -- the geometry of a scene behind its first grid pooling, enqueued by one native call.
//
// GridPool's clustering (point_transformer_v2m2_base.py:246-268), the self k-NN tables of the pooled levels (:223), the 3-NN
// interpolation tables back to the finer level (:311) and the inverse tables the fixed-order backward gathers need depend on
// the coordinates only.  ao_amd/ptv2/geometry.py builds them with one python-level launcher call per table -- ~25 calls,
// ~2 ms of host time per 120 k-point scene (tensor allocations, ctypes marshalling, workspace look-ups), all of it on the
// critical path of a forward that builds its own geometry (ao_amd/ptv2/native_model.py: the host has the ~1.7 ms the
// level-0 prefix of the network computes to get the rest of the network enqueued).  Here the same launchers are called back to
// back from native code; the only host waits left are the three 4-byte read-backs of the poolings' cluster counts, which size
// the next level's tables -- carved from the caller's arena as the sizes become known.
#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

constexpr int KMAX = 32;  // largest K of a self table handled here (the grid query's limit; PT-v2m2 uses 8 and 16)

struct Ws {
    char *knn; size_t knn_bytes;      // the cell grid of the level being queried + the queries' scratch
    char *pool; size_t pool_bytes;
    char *pos; size_t pos_bytes;
    char *inv; size_t inv_bytes;
    float *dist2;                     // (n0, KMAX)
    int *n_out;
    size_t bytes;
};

int count_tables(const ptv2_scene_geo *G) {
    int t = 0;
    for (int i = 0; i <= G->num_stages; ++i) t += G->level[i].nk;
    return t + (G->interp ? G->num_stages : 0);
}

bool geo_ok(const ptv2_scene_geo *G) {
    if (!G || G->num_stages < 1 || G->num_stages > PTV2_MAX_STAGES || G->b < 1 || G->level[0].n < 1) return false;
    if (!G->coord0 || !G->offset0) return false;
    for (int i = 0; i <= G->num_stages; ++i) {
        const ptv2_geo_level &L = G->level[i];
        if (L.nk < 0 || L.nk > PTV2_GEO_MAX_K) return false;
        for (int j = 0; j < L.nk; ++j)
            if (L.knn[j].k < 1 || L.knn[j].k > KMAX) return false;
    }
    for (int j = 0; j < G->level[0].nk; ++j)
        if (!G->knn0[j]) return false;
    for (int i = 0; i < G->num_stages; ++i)
        if (!(G->grid_size[i] > 0.f)) return false;
    return true;
}

Ws carve_ws(const ptv2_scene_geo *G, void *base) {
    Ws w;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += al(bytes); return r; };
    const int n0 = G->level[0].n, b = G->b;
    // every level has at most n0 points: the bounds below hold for all of them (checked again at run time)
    w.knn_bytes = knn_query_hip_workspace_bytes(n0, n0, b);
    w.knn = take(w.knn_bytes);
    w.pool_bytes = grid_pool_hip_workspace_bytes(n0, b);
    w.pool = take(w.pool_bytes);
    w.pos_bytes = gva_workspace_bytes(n0, KMAX, 8, 1);
    w.pos = take(w.pos_bytes);
    {
        ptv2_inverse_job jobs[PTV2_INVERSE_MAX_JOBS];
        const int cnt = std::min(count_tables(G), (int)PTV2_INVERSE_MAX_JOBS);
        for (int j = 0; j < cnt; ++j) { jobs[j].n = n0; jobs[j].k = KMAX; jobs[j].idx = nullptr; jobs[j].inv_ptr = nullptr; jobs[j].inv_rows = nullptr; }
        w.inv_bytes = cnt ? inverse_tables_hip_workspace_bytes(cnt, jobs) : 0;
    }
    w.inv = take(w.inv_bytes);
    w.dist2 = (float *)take(sizeof(float) * (size_t)n0 * KMAX);
    w.n_out = (int *)take(256);
    w.bytes = off;
    return w;
}

// bytes of the tables of one level with n rows whose finer / coarser neighbours have at most n rows as well
size_t level_bytes(const ptv2_scene_geo *G, int i, size_t n, bool first, bool last) {
    size_t t = 0;
    const ptv2_geo_level &L = G->level[i];
    if (!first) t += al(12 * n) + al(4 * (size_t)G->b);
    for (int j = 0; j < L.nk; ++j) {
        const size_t k = L.knn[j].k;
        if (!first) t += al(4 * n * k) + al(24) + al(72);
        t += al(4 * (n + 1)) + al(4 * n * k);
    }
    if (!last) {
        t += al(8 * n) + al(4 * n) + al(4 * (n + 1));
        if (G->interp) t += al(12 * n) + al(12 * n) + al(4 * (n + 1)) + al(12 * n);
    }
    return t;
}

// Pinned read-back words, one per call in flight (a pool: the mutex covers taking / returning a word only -- a geometry build
// that waits for its GPU stage no longer keeps every other build of the process, on whatever device or stream, waiting behind
// it; hipHostMallocPortable: the word is usable from every device of the process)
std::mutex g_pin_mu;
std::vector<int *> g_pin_free;
struct PinnedWord {
    int *p = nullptr;
    PinnedWord() {
        {
            std::lock_guard<std::mutex> lk(g_pin_mu);
            if (!g_pin_free.empty()) { p = g_pin_free.back(); g_pin_free.pop_back(); }
        }
        if (!p && hipHostMalloc((void **)&p, 64, hipHostMallocPortable) != hipSuccess) p = nullptr;
    }
    ~PinnedWord() {
        if (!p) return;
        std::lock_guard<std::mutex> lk(g_pin_mu);
        g_pin_free.push_back(p);
    }
    PinnedWord(const PinnedWord &) = delete;
    PinnedWord &operator=(const PinnedWord &) = delete;
};

}  // namespace

extern "C" size_t ptv2_scene_geometry_arena_bytes(const ptv2_scene_geo *G) {
    if (!geo_ok(G)) return 0;
    size_t t = 256;
    const size_t n0 = G->level[0].n;
    for (int i = 0; i <= G->num_stages; ++i) t += level_bytes(G, i, n0, i == 0, i == G->num_stages);
    return t;
}

extern "C" size_t ptv2_scene_geometry_workspace_bytes(const ptv2_scene_geo *G) {
    if (!geo_ok(G)) return 0;
    return carve_ws(G, nullptr).bytes + 256;
}

#define RUN(call)                        \
    do {                                 \
        int rc_ = (call);                \
        if (rc_ != PTV2_OK) return rc_;  \
    } while (0)

static int scene_geometry(ptv2_scene_geo *G, void *arena, size_t arena_bytes, void *workspace, size_t workspace_bytes, void *stream);

extern "C" int ptv2_scene_geometry_hip_launcher(ptv2_scene_geo *G, void *arena, size_t arena_bytes, void *workspace,
                                                size_t workspace_bytes, void *stream) {
    if (!G) return PTV2_ERR_ARG;
    G->sizes_ready = 0;
    G->fwd_recorded = 0;
    const int rc = scene_geometry(G, arena, arena_bytes, workspace, workspace_bytes, stream);
    if (rc != PTV2_OK) __atomic_store_n(&G->sizes_ready, -1, __ATOMIC_RELEASE);  // (a polling thread must not wait for ever)
    return rc;
}

static int scene_geometry(ptv2_scene_geo *G, void *arena, size_t arena_bytes, void *workspace, size_t workspace_bytes, void *stream) {
    if (!geo_ok(G) || !arena) return PTV2_ERR_ARG;
    const Ws W = carve_ws(G, workspace);
    if (!workspace || workspace_bytes < W.bytes) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int S = G->num_stages, b = G->b;
    char *base = (char *)arena;
    size_t off = 0;
    bool full = false;
    auto take = [&](size_t bytes) -> long long {
        const size_t at = off;
        off += al(bytes);
        if (off > arena_bytes) { full = true; return -1; }
        return (long long)at;
    };
    auto at = [&](long long o) { return base + o; };
    for (int i = 0; i <= S; ++i) {  // nothing produced yet
        ptv2_geo_level &L = G->level[i];
        L.coord = L.offset = L.cluster = L.order = L.idx_ptr = L.up_idx = L.up_w = L.up_inv_ptr = L.up_inv_rows = -1;
        for (int j = 0; j < PTV2_GEO_MAX_K; ++j) L.knn[j].idx = L.knn[j].mu = L.knn[j].cov = L.knn[j].inv_ptr = L.knn[j].inv_rows = -1;
    }
    const PinnedWord pinned;
    if (!pinned.p) return PTV2_ERR_LAUNCH;
    int *const g_pinned = pinned.p;

    // ---- phase 1: the chain of grid poolings alone -- each needs only the coordinates of the level above it, and their cluster
    // counts are the only data-dependent sizes of a scene.  With all S read-backs at the front, a second host thread (the one
    // that issues the network: ao_amd/ptv2/native_model.py) learns every size ~0.3 ms into the call and prepares the rest of the
    // forward while the tables below are still being computed.
    const float *coords[PTV2_MAX_STAGES + 1];
    const int *offsets[PTV2_MAX_STAGES + 1];
    coords[0] = G->coord0;
    offsets[0] = G->offset0;
    for (int i = 0; i < S; ++i) {
        ptv2_geo_level &L = G->level[i], &N = G->level[i + 1];
        const int n = L.n;
        // GridPool's clustering of level i: cluster / order exact, idx_ptr / pooled coordinates for at most n clusters
        L.cluster = take(8 * (size_t)n);
        L.order = take(4 * (size_t)n);
        L.idx_ptr = take(4 * ((size_t)n + 1));
        N.offset = take(4 * (size_t)b);
        N.coord = take(12 * (size_t)n);
        if (full) return PTV2_ERR_WORKSPACE;
        if (grid_pool_hip_workspace_bytes(n, b) > W.pool_bytes) return PTV2_ERR_WORKSPACE;
        int m = -2;
        for (int sort_path = 0; sort_path < 2 && m == -2; ++sort_path) {  // the dense table first; -2 asks for the radix sort
            RUN(grid_pool_hip_launcher(n, b, coords[i], offsets[i], G->grid_size[i], (long long *)at(L.cluster), (int *)at(L.order),
                                       (int *)at(L.idx_ptr), (float *)at(N.coord), (int *)at(N.offset), W.n_out, sort_path, W.pool,
                                       W.pool_bytes, stream));
            if (hipMemcpyAsync(g_pinned, W.n_out, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return PTV2_ERR_LAUNCH;
            if (hipStreamSynchronize(st) != hipSuccess) return PTV2_ERR_LAUNCH;  // the one host wait of a stage
            m = *g_pinned;
        }
        if (m < 1) {  // voxel ids beyond the sort key (m == -1, reported as such: the caller tells it from other argument errors)
            N.n = m == -1 ? -1 : 0;
            return PTV2_ERR_ARG;
        }
        N.n = m;
        off = (size_t)N.coord + al(12 * (size_t)m);  // the pooled coordinates were the last item: give the unused rows back
        coords[i + 1] = (const float *)at(N.coord);
        offsets[i + 1] = (const int *)at(N.offset);
        if (knn_query_hip_workspace_bytes(std::max(n, m), m, b) > W.knn_bytes) return PTV2_ERR_WORKSPACE;
    }
    // ---- every size is known: carve all the tables; from here on every n and offset of the struct is final
    for (int i = 0; i <= S; ++i) {
        ptv2_geo_level &L = G->level[i];
        if (i < S && G->interp) {
            L.up_idx = take(12 * (size_t)L.n);
            L.up_w = take(12 * (size_t)L.n);
        }
        if (i > 0)
            for (int j = 0; j < L.nk; ++j) {
                ptv2_geo_table &T = L.knn[j];
                T.idx = take(4 * (size_t)L.n * T.k);
                T.mu = take(24);
                T.cov = take(72);
            }
    }
    for (int l = 0; l <= S; ++l) {
        ptv2_geo_level &Q = G->level[l];
        for (int j = 0; j < Q.nk; ++j) {
            Q.knn[j].inv_ptr = take(4 * ((size_t)Q.n + 1));
            Q.knn[j].inv_rows = take(4 * (size_t)Q.n * Q.knn[j].k);
        }
        if (l < S && G->interp) {
            Q.up_inv_ptr = take(4 * ((size_t)Q.n + 1));
            Q.up_inv_rows = take(4 * (size_t)Q.n * 3);
        }
    }
    if (full) return PTV2_ERR_WORKSPACE;
    __atomic_store_n(&G->sizes_ready, 1, __ATOMIC_RELEASE);
    // ---- phase 2: the tables over the points of every pooled level; ONE cell grid per level serves the interpolation query from
    // the finer level and the level's self tables
    for (int i = 0; i < S; ++i) {
        ptv2_geo_level &L = G->level[i], &N = G->level[i + 1];
        const int n = L.n, m = N.n;
        const float *coord = coords[i], *ncoord = coords[i + 1];
        const int *offset = offsets[i], *noffset = offsets[i + 1];
        int used = 0;
        if (G->interp) {
            RUN(knn_query_grid_hip_launcher(n, 3, ncoord, coord, noffset, offset, (int *)at(L.up_idx), W.dist2, m, b, 0, 0, 0, W.knn,
                                            W.knn_bytes, stream));
            used = 1;
            RUN(interpolation_weights_hip_launcher(n, 3, m, W.dist2, (int *)at(L.up_idx), (float *)at(L.up_w), stream));
        }
        for (int j = 0; j < N.nk; ++j) {
            ptv2_geo_table &T = N.knn[j];
            if (used > 3) used = 0;  // (more queries than re-run counters: rebuild)
            RUN(knn_query_grid_hip_launcher(m, T.k, ncoord, ncoord, noffset, noffset, (int *)at(T.idx), W.dist2, m, b, 0, used ? 1 : 0,
                                            used, W.knn, W.knn_bytes, stream));
            ++used;
            RUN(gva_pos_moments_hip_launcher(m, T.k, ncoord, (const int *)at(T.idx), (double *)at(T.mu), (double *)at(T.cov), W.pos,
                                             W.pos_bytes, stream));
        }
    }
    if (G->fwd_ready_event && hipEventRecord((hipEvent_t)G->fwd_ready_event, st) != hipSuccess) return PTV2_ERR_LAUNCH;
    __atomic_store_n(&G->fwd_recorded, 1, __ATOMIC_RELEASE);
    if (G->knn0_event && hipStreamWaitEvent(st, (hipEvent_t)G->knn0_event, 0) != hipSuccess) return PTV2_ERR_LAUNCH;
    // ---- inverse tables of every table of the scene, PTV2_INVERSE_MAX_JOBS per call (four launches each)
    ptv2_inverse_job jobs[PTV2_INVERSE_MAX_JOBS];
    int cnt = 0;
    auto flush = [&]() -> int {
        if (!cnt) return PTV2_OK;
        if (inverse_tables_hip_workspace_bytes(cnt, jobs) > W.inv_bytes) return PTV2_ERR_WORKSPACE;
        const int rc = inverse_tables_hip_launcher(cnt, jobs, W.inv, W.inv_bytes, stream);
        cnt = 0;
        return rc;
    };
    auto add = [&](int rows, int k, const int *idx, int targets, long long *inv_ptr, long long *inv_rows) -> int {
        (void)targets;  // (carved behind the last read-back, above)
        ptv2_inverse_job &J = jobs[cnt++];
        J.n = rows; J.k = k; J.idx = idx; J.inv_ptr = (int *)at(*inv_ptr); J.inv_rows = (int *)at(*inv_rows);
        if (cnt == PTV2_INVERSE_MAX_JOBS) return flush();
        return PTV2_OK;
    };
    for (int i = 0; i <= S; ++i) {
        ptv2_geo_level &L = G->level[i];
        for (int j = 0; j < L.nk; ++j) {
            const int *idx = i == 0 ? G->knn0[j] : (const int *)at(L.knn[j].idx);
            RUN(add(L.n, L.knn[j].k, idx, L.n, &L.knn[j].inv_ptr, &L.knn[j].inv_rows));
        }
        if (i < S && G->interp) {
            // (n_i, 3) entries into level i + 1.  The inverse-table launcher takes tables whose targets are its own rows
            // (entries in [-1, n)): the rows n_{i+1} .. n_i - 1 of inv_ptr repeat the end -- n_{i+1} <= n_i
            RUN(add(L.n, 3, (const int *)at(L.up_idx), L.n, &L.up_inv_ptr, &L.up_inv_rows));
        }
    }
    RUN(flush());
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
