"""Scene geometry for PT-v2m2: everything that depends only on (coord, offset).

In the reference the geometric work is interleaved with the network: every BlockSequence calls
knn_query (point_transformer_v2m2_base.py:223, recomputed in the decoder on identical coords),
every GridPool voxelises + sorts (:244-269, with a host-sync python loop in offset2batch), every
"interp" unpool runs a 3-NN (:311).  None of it depends on features or weights, so here it is
computed ONCE per batch, up front, under no_grad: 1 + S self-kNNs, S grid poolings, S cross-kNNs
for an S-stage model -- and handed to the layers as a `SceneGeometry`.
"""
import ctypes
import os
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .. import _lib, pointops
from ..pointops.interpolation import interpolation_index_weight
from ..pointops.query import KnnGrid, knn_query_dist2


@dataclass
class Level:
    coord: torch.Tensor            # (N,3) fp32
    offset: torch.Tensor           # (B,) int32 cumulative
    knn: dict = field(default_factory=dict)  # K -> (N,K) int32 neighbour table of this level
    # link to the next coarser level (filled for all but the last level)
    cluster: Optional[torch.Tensor] = None   # (N,) int64: fine point -> coarse point
    order32: Optional[torch.Tensor] = None   # (N,) int32: fine points sorted by cluster (stable)
    idx_ptr32: Optional[torch.Tensor] = None  # (N'+1,) int32 CSR over `order32`
    # link from the next coarser level back to this one ("interp" unpooling)
    up_idx: Optional[torch.Tensor] = None    # (N,3) int32 into the coarser level
    up_weight: Optional[torch.Tensor] = None  # (N,3) fp32
    grid: Optional[object] = None            # pointops.query.KnnGrid while the scene is being built (one cell grid per level)


    def neighbours(self, k, inverse=True):
        """Self k-NN table of this level; one kNN launch per distinct K, shared by every BlockSequence
        that works at this resolution (encoder and decoder).  inverse=False: the caller builds the inverse tables of all
        its levels in one call afterwards (gva.inverse_tables)."""
        if k not in self.knn:
            with torch.no_grad():
                idx = knn_query_dist2(k, self.coord, self.offset, grid=self.grid)[0]  # (the table only: no sqrt of the distances)
                # table-only quantities the fused attention needs (their host syncs belong to the geometry phase)
                from . import gva
                if gva.supported(8 * 6, 6, k):
                    if inverse:
                        gva.inverse_table(idx)
                    gva._pos_moments(gva._HipImpl, self.coord, idx)
                self.knn[k] = idx
        return self.knn[k]


@dataclass
class SceneGeometry:
    levels: List[Level] = field(default_factory=list)


def offset2batch(offset):
    off = offset.long()
    counts = torch.diff(off, prepend=off.new_zeros(1))
    return torch.repeat_interleave(torch.arange(off.numel(), device=off.device), counts)


def segment_minmax(coord, offset):
    """Per-cloud coordinate (min, max), each (B,3) (ao_amd/csrc/pool.hip)."""
    _lib.require_cuda(coord, offset)
    b = offset.numel()
    lo = torch.empty((b, 3), dtype=torch.float32, device=coord.device)
    hi = torch.empty((b, 3), dtype=torch.float32, device=coord.device)
    L = _lib.lib()
    ws = _lib.workspace(L.segment_minmax_hip_workspace_bytes(b), coord.device)
    rc = L.segment_minmax_hip_launcher(b, coord.data_ptr(), offset.int().contiguous().data_ptr(), lo.data_ptr(),
                                       hi.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "segment_minmax_hip_launcher")
    return lo, hi


def voxel_cluster_ids(coord, offset, grid_size, start=None):
    """Cluster id of GridPool (:246-259): per-cloud min-shifted coords (or the caller's `start`, (B,3)),
    torch_cluster.grid_cluster formula with the batch index as the most significant digit
    (oracle/ptv2_ref.py:voxel_grid)."""
    batch = offset2batch(offset)
    if start is None:
        start, _ = segment_minmax(coord, offset)
    pos = coord - start[batch]
    size = coord.new_tensor([grid_size, grid_size, grid_size])
    num = (pos.max(0)[0] / size).long() + 1          # voxels per axis over the whole batch
    cell = (pos / size.view(1, 3)).long()
    stride_y, stride_z, stride_b = num[0], num[0] * num[1], num[0] * num[1] * num[2]
    return cell[:, 0] + cell[:, 1] * stride_y + cell[:, 2] * stride_z + batch * stride_b, batch


def grid_pool_geometry_torch(coord, offset, grid_size, start=None):
    """Coordinates-only half of GridPool.forward (:257-268) as torch ops (kept as the in-framework statement
    of what the device kernel computes; AO_AMD_GRIDPOOL=torch selects it)."""
    key, batch = voxel_cluster_ids(coord, offset, grid_size, start)
    _, cluster, counts = torch.unique(key, sorted=True, return_inverse=True, return_counts=True)
    order = torch.sort(cluster, stable=True)[1]
    idx_ptr = torch.cat([counts.new_zeros(1), torch.cumsum(counts, dim=0)])
    new_coord = torch.segment_reduce(coord[order], "mean", offsets=idx_ptr, axis=0)
    new_batch = batch[order[idx_ptr[:-1]]]
    new_offset = torch.cumsum(new_batch.bincount(minlength=offset.numel()), dim=0).int()
    return new_coord.contiguous(), new_offset, cluster, order.int().contiguous(), idx_ptr.int().contiguous()


def grid_pool_geometry(coord, offset, grid_size):
    """(new_coord (N',3), new_offset (B) int32, cluster (N) int64, order (N) int32, idx_ptr (N'+1) int32) on
    ao_amd/csrc/gridpool.hip: one launcher call and ONE 4-byte read-back (the number of clusters)."""
    if os.environ.get("AO_AMD_GRIDPOOL", "hip") == "torch":
        return grid_pool_geometry_torch(coord, offset, grid_size)
    _lib.require_cuda(coord, offset)
    n, b = coord.shape[0], offset.numel()
    dev = coord.device
    off = offset.int().contiguous()
    cluster = torch.empty(n, dtype=torch.int64, device=dev)
    order = torch.empty(n, dtype=torch.int32, device=dev)
    idx_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    new_coord = torch.empty((n, 3), dtype=torch.float32, device=dev)
    new_offset = torch.empty(b, dtype=torch.int32, device=dev)
    n_out = torch.empty(1, dtype=torch.int32, device=dev)
    L = _lib.lib()
    ws = _lib.workspace(L.grid_pool_hip_workspace_bytes(n, b), dev)
    def launch(sort_path):
        rc = L.grid_pool_hip_launcher(n, b, coord.data_ptr(), off.data_ptr(), float(grid_size), cluster.data_ptr(),
                                      order.data_ptr(), idx_ptr.data_ptr(), new_coord.data_ptr(), new_offset.data_ptr(),
                                      n_out.data_ptr(), sort_path, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "grid_pool_hip_launcher")
        return int(n_out.item())  # the one host sync of the pooling: the output size is data dependent

    # the voxel grid is tabulated (no sort) when it is small enough -- which only the device knows: -2 asks for the sort path
    m = launch(1 if os.environ.get("AO_AMD_GRIDPOOL") == "sort" else 0)
    if m == -2:
        m = launch(1)
    if m < 0:
        raise RuntimeError("grid_pool: voxel ids exceed the 48-bit sort key (scene extent / grid_size too large)")
    return new_coord[:m], new_offset, cluster, order, idx_ptr[: m + 1]


@torch.no_grad()
def build_geometry(coord, offset, grid_sizes, neighbours, interp=True):
    """neighbours[i] = iterable of K values needed at level i (level 0 = input resolution,
    level i+1 = after grid_sizes[i])."""
    state = begin_geometry(coord, offset, grid_sizes, neighbours, interp)
    if native_finish_supported(state):  # one native call for everything behind the first pooling (csrc/scene.hip)
        return finish_geometry_native(state)
    return finish_geometry(state)


class _GeometryState:
    __slots__ = ("geo", "cur", "grid_sizes", "neighbours", "interp")


@torch.no_grad()
def begin_geometry(coord, offset, grid_sizes, neighbours, interp=True):
    """First half of build_geometry: the level-0 neighbour tables (+ their position moments) -- everything the network's
    level-0 prefix (patch embedding) needs, and nothing whose size is data dependent.  Returns the state finish_geometry
    continues from; `state.geo.levels[0]` is usable at once (on the stream this ran on)."""
    st = _GeometryState()
    st.geo = SceneGeometry()
    # one cell grid per level, shared by the queries over its points: the interpolation table from the finer level (k = 3) and
    # the level's self tables (csrc/knn.hip: knn_query_grid_hip_launcher)
    st.cur = Level(coord=coord.contiguous(), offset=offset.int().contiguous(), grid=KnnGrid())
    st.grid_sizes, st.neighbours, st.interp = list(grid_sizes), [list(ks) for ks in neighbours], interp
    for k in st.neighbours[0]:
        st.cur.neighbours(k, inverse=False)
    st.geo.levels.append(st.cur)
    return st


@torch.no_grad()
def finish_geometry(st, before_inverse=None):
    """Second half: the grid poolings (one 4-byte read-back each: the next level's size), the deeper levels' tables, the
    interpolation tables and the inverse tables of the whole scene.  May run on another stream than begin_geometry did
    (native_model's pipelined forward: a side stream, while the level-0 prefix computes); `before_inverse()` is called in
    front of the one launch that reads the level-0 tables (the caller's cross-stream wait)."""
    geo, cur = st.geo, st.cur
    for i, ks in enumerate(st.neighbours):
        if i > 0:
            for k in ks:
                cur.neighbours(k, inverse=False)
            geo.levels.append(cur)
        if i == len(st.grid_sizes):
            break
        nc, noff, cluster, order, idx_ptr = grid_pool_geometry(cur.coord, cur.offset, st.grid_sizes[i])
        cur.cluster, cur.order32, cur.idx_ptr32 = cluster, order, idx_ptr
        nxt = Level(coord=nc, offset=noff, grid=KnnGrid())
        if st.interp:
            cur.up_idx, cur.up_weight = interpolation_index_weight(nc, cur.coord, noff, cur.offset, 3, grid=nxt.grid)
        cur = nxt
    # inverse tables of every neighbour / interpolation table of the scene in one call (five launches: csrc/inverse.hip);
    # they let the backward gather in a fixed order instead of scattering with float atomics
    from . import gva
    tables = [lv.up_idx for lv in geo.levels if lv.up_idx is not None]
    tables += [idx for lv in geo.levels for k, idx in lv.knn.items() if gva.supported(8 * 6, 6, k)]
    if before_inverse is not None:
        before_inverse()
    with torch.no_grad():
        gva.inverse_tables(tables)
    for lv in geo.levels:
        lv.grid = None  # (the grids' workspaces go back to the allocator; a table asked for later builds its own)
    return geo


# ---- the second half as ONE native call (ao_amd/csrc/scene.hip) --------------------------------------------------------------
_MAX_STAGES, _GEO_MAX_K = 5, 2
_LL = ctypes.c_longlong


class _GeoTable(ctypes.Structure):  # mirrors ptv2_geo_table
    _fields_ = [("k", ctypes.c_int)] + [(n, _LL) for n in ("idx", "mu", "cov", "inv_ptr", "inv_rows")]


class _GeoLevel(ctypes.Structure):  # mirrors ptv2_geo_level
    _fields_ = ([("n", ctypes.c_int), ("nk", ctypes.c_int), ("knn", _GeoTable * _GEO_MAX_K)]
                + [(n, _LL) for n in ("coord", "offset", "cluster", "order", "idx_ptr", "up_idx", "up_w", "up_inv_ptr", "up_inv_rows")])


class _SceneGeo(ctypes.Structure):  # mirrors ptv2_scene_geo
    _fields_ = [("num_stages", ctypes.c_int), ("b", ctypes.c_int), ("interp", ctypes.c_int), ("grid_size", ctypes.c_float * _MAX_STAGES),
                ("coord0", ctypes.c_void_p), ("offset0", ctypes.c_void_p), ("knn0", ctypes.c_void_p * _GEO_MAX_K),
                ("fwd_ready_event", ctypes.c_void_p), ("knn0_event", ctypes.c_void_p), ("level", _GeoLevel * (_MAX_STAGES + 1)),
                ("sizes_ready", ctypes.c_int), ("fwd_recorded", ctypes.c_int)]


_lib.register({
    "ptv2_scene_geometry_arena_bytes": (_lib._c_size, [ctypes.c_void_p]),
    "ptv2_scene_geometry_workspace_bytes": (_lib._c_size, [ctypes.c_void_p]),
    "ptv2_scene_geometry_hip_launcher": (_lib._c_int, [ctypes.c_void_p, ctypes.c_void_p, _lib._c_size, ctypes.c_void_p, _lib._c_size,
                                                       ctypes.c_void_p]),
})
_lib.check_struct(4, _SceneGeo)


def native_finish_supported(st):
    return (len(st.grid_sizes) <= _MAX_STAGES and all(1 <= len(ks) <= _GEO_MAX_K and all(1 <= k <= 32 for k in ks) for ks in st.neighbours)
            and len(st.neighbours) == len(st.grid_sizes) + 1 and os.environ.get("AO_AMD_GEOMETRY", "native") == "native"
            and os.environ.get("AO_AMD_GRIDPOOL", "hip") == "hip")


class NativeGeometryJob:
    """One ptv2_scene_geometry_hip_launcher call, split so that the native call itself can run on a helper thread: `prepare`
    (struct, arena, workspace: on the thread and stream the tensors are to be allocated on), `run` (the ctypes call alone --
    it releases the interpreter lock and blocks in the poolings' read-backs), `wait_sizes` / `wait_fwd_recorded` (another
    thread polls the struct's progress flags), `geometry` (the RawSceneGeometry once the sizes are in)."""

    def __init__(self, st, fwd_ready_event=None, knn0_event=None):
        lv0 = st.cur
        dev = lv0.coord.device
        S = len(st.grid_sizes)
        G = _SceneGeo()
        G.num_stages, G.b, G.interp = S, lv0.offset.numel(), 1 if st.interp else 0
        for i, g in enumerate(st.grid_sizes):
            G.grid_size[i] = float(g)
        G.coord0, G.offset0 = lv0.coord.data_ptr(), lv0.offset.data_ptr()
        G.level[0].n = lv0.coord.shape[0]
        for i, ks in enumerate(st.neighbours):
            G.level[i].nk = len(ks)
            for j, k in enumerate(ks):
                G.level[i].knn[j].k = int(k)
        for j, k in enumerate(st.neighbours[0]):
            G.knn0[j] = lv0.knn[k].data_ptr()
        G.fwd_ready_event = fwd_ready_event.cuda_event if fwd_ready_event is not None else None
        G.knn0_event = knn0_event.cuda_event if knn0_event is not None else None
        L = _lib.lib()
        addr = ctypes.addressof(G)
        self.st, self.G, self.lv0, self.L = st, G, lv0, L
        self.arena = torch.empty(L.ptv2_scene_geometry_arena_bytes(addr), dtype=torch.uint8, device=dev)
        self.ws = _lib.workspace(L.ptv2_scene_geometry_workspace_bytes(addr), dev)
        self.stream = _lib.stream_ptr()
        self.rc = None

    def run(self):
        G = self.G
        self.rc = self.L.ptv2_scene_geometry_hip_launcher(ctypes.addressof(G), self.arena.data_ptr(), self.arena.numel(),
                                                          self.ws.data_ptr(), self.ws.numel(), self.stream)
        return self.rc

    def check(self):
        rc, G = self.rc, self.G
        if rc == 1 and any(G.level[i + 1].n == -1 for i in range(G.num_stages)):  # (the launcher marks exactly this case)
            raise RuntimeError("grid_pool: voxel ids exceed the 48-bit sort key (scene extent / grid_size too large)")
        _lib.check(rc, "ptv2_scene_geometry_hip_launcher")

    def _poll(self, field, fut):
        import time
        G = self.G
        while getattr(G, field) == 0:
            if fut is not None and fut.done():
                break
            time.sleep(0)  # (hands the interpreter lock over; the helper thread is inside the native call almost all the time)
        if getattr(G, field) != 1:
            if fut is not None:
                fut.result()
            self.check()
            raise RuntimeError("ptv2_scene_geometry_hip_launcher ended without reporting its sizes")

    def wait_sizes(self, fut=None):
        self._poll("sizes_ready", fut)

    def wait_fwd_recorded(self, fut=None):
        self._poll("fwd_recorded", fut)

    def geometry(self):
        return RawSceneGeometry(self.st, self.G, self.arena, self.lv0)


@torch.no_grad()
def finish_geometry_native(st, fwd_ready_event=None, knn0_event=None):
    """finish_geometry as one native call (ptv2_scene_geometry_hip_launcher): the same launchers in the same order, enqueued
    from native code; the tables of levels 1.. live in ONE arena tensor, carved as the poolings' sizes are read back.
    Runs on the current stream (and synchronises it once per stage).  The events are torch.cuda.Event objects that have been
    recorded at least once (their handle exists): `fwd_ready_event` is recorded once everything the forward needs is enqueued,
    `knn0_event` is waited for in front of the inverse tables."""
    job = NativeGeometryJob(st, fwd_ready_event, knn0_event)
    job.run()
    job.check()
    return job.geometry()


class RawSceneGeometry(SceneGeometry):
    """The result of ptv2_scene_geometry_hip_launcher: level 0's tensors, one arena tensor and the struct of byte offsets into
    it.  The native model runtime takes sizes and addresses straight from here (native_model._Runtime.fill_geometry: no tensor
    objects on the launching thread's critical path -- ~60 views cost 0.3 ms of host time per step); `levels`, the list of
    Level objects every other consumer reads, is built on first use."""

    def __init__(self, st, G, arena, lv0):
        self.G, self.arena, self.lv0 = G, arena, lv0
        self.neighbours, self.interp = st.neighbours, st.interp
        self.sizes = [G.level[i].n for i in range(G.num_stages + 1)]
        self.base = arena.data_ptr()
        self._levels = None

    def addr(self, off):
        return self.base + off if off >= 0 else None

    def table(self, level, k):
        """(idx, mu, cov, inv_ptr, inv_rows) addresses of the level's K = k self table."""
        j = self.neighbours[level].index(k)
        T = self.G.level[level].knn[j]
        if level == 0:
            idx = self.lv0.knn[k]
            _, mu, cov = idx._ao_pos_moments
            return idx.data_ptr(), mu.data_ptr(), cov.data_ptr(), self.addr(T.inv_ptr), self.addr(T.inv_rows)
        return self.addr(T.idx), self.addr(T.mu), self.addr(T.cov), self.addr(T.inv_ptr), self.addr(T.inv_rows)

    def tensors(self):
        """what has to stay alive (and be marked for the consuming stream) while kernels read the geometry"""
        out = [self.arena, self.lv0.coord, self.lv0.offset]
        for idx in self.lv0.knn.values():
            out.append(idx)
            out.extend(t for t in getattr(idx, "_ao_pos_moments", ())[1:] if torch.is_tensor(t))
        return out

    @property
    def levels(self):
        if self._levels is None:
            self._levels = self._build_levels()
        return self._levels

    @levels.setter
    def levels(self, value):  # (SceneGeometry's dataclass __init__ is not used; kept assignable)
        self._levels = value

    def _build_levels(self):
        G, arena, lv0, S = self.G, self.arena, self.lv0, self.G.num_stages

        def view(off, dtype, *shape):
            if off < 0:
                return None
            count = 1
            for d in shape:
                count *= d
            nbytes = count * torch.empty((), dtype=dtype).element_size()
            return arena[off:off + nbytes].view(dtype).view(*shape)

        def attach(idx, coord, T):
            inv_ptr, inv_rows = view(T.inv_ptr, torch.int32, idx.shape[0] + 1), view(T.inv_rows, torch.int32, idx.numel())
            idx._ao_inverse = (idx._version, inv_ptr, inv_rows)
            if T.mu >= 0:
                idx._ao_pos_moments = ((coord.data_ptr(), idx._version), view(T.mu, torch.float64, 3), view(T.cov, torch.float64, 3, 3))

        b = G.b
        levels = [lv0]
        for i in range(1, S + 1):
            Li = G.level[i]
            lv = Level(coord=view(Li.coord, torch.float32, Li.n, 3), offset=view(Li.offset, torch.int32, b))
            for j, k in enumerate(self.neighbours[i]):
                lv.knn[k] = view(Li.knn[j].idx, torch.int32, Li.n, k)
                attach(lv.knn[k], lv.coord, Li.knn[j])
            levels.append(lv)
        for j, k in enumerate(self.neighbours[0]):
            attach(lv0.knn[k], lv0.coord, G.level[0].knn[j])
        for i in range(S):
            Li, lv = G.level[i], levels[i]
            n, m = Li.n, G.level[i + 1].n
            lv.cluster, lv.order32 = view(Li.cluster, torch.int64, n), view(Li.order, torch.int32, n)
            lv.idx_ptr32 = view(Li.idx_ptr, torch.int32, m + 1)
            if self.interp:
                lv.up_idx, lv.up_weight = view(Li.up_idx, torch.int32, n, 3), view(Li.up_w, torch.float32, n, 3)
                lv.up_idx._ao_inverse = (lv.up_idx._version, view(Li.up_inv_ptr, torch.int32, n + 1), view(Li.up_inv_rows, torch.int32, 3 * n))
        for lv in levels:
            lv.grid = None
        return levels
