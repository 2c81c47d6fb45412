"""k-NN query on the HIP grid kernel (ao_amd/csrc/knn.hip).

Mirrors libs/pointops/functions/query.py:7-24,111: same name, argument order, dtypes, -1 / 1e10
placeholders and the sqrt on the way out.  ball_query / random_ball_query are outside the
PT-v2m2 path and raise.
"""
import torch
from torch.autograd import Function

from .. import _lib


class KnnGrid:
    """The cell grid of one set of source points (xyz, offset), kept in a workspace of its own so that several queries over
    the same points -- the interpolation table from the finer level and the self tables of a level -- build it once
    (knn_query_grid_hip_launcher).  Valid on the stream it was built on, as long as xyz / offset are not written."""
    MAX_QUERIES = 4  # re-run counters the workspace carries

    def __init__(self):
        self.key, self.ws, self.used = None, None, 0

    def plan(self, xyz, off, m, n, b, nbytes, stream):
        """(grid_mode, slot, workspace) for the next query: reuse when the grid in the workspace is this one."""
        key = (xyz.data_ptr(), off.data_ptr(), n, b, stream)
        if self.key == key and self.ws is not None and self.ws.numel() >= nbytes and self.used < self.MAX_QUERIES:
            self.used += 1
            return 1, self.used - 1, self.ws
        if self.ws is None or self.ws.numel() < nbytes:
            self.ws = torch.empty(int(nbytes) + 4096, dtype=torch.uint8, device=xyz.device)
        self.key, self.used, self.keep = key, 1, (xyz, off)
        return 0, 0, self.ws


def knn_query_dist2(nsample, xyz, offset, new_xyz=None, new_offset=None, pad_with_start=False, grid=None):
    """(idx int32 (m,k), squared distances fp32 (m,k)) -- the raw kernel outputs.  grid: a KnnGrid shared by the queries over
    the same (xyz, offset) (pass the SAME offset tensor to each of them)."""
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    _lib.require_cuda(xyz, new_xyz, offset, new_offset)
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    assert xyz.dtype == torch.float32 and new_xyz.dtype == torch.float32
    self_query = new_xyz is xyz and new_offset is offset
    off = offset if offset.dtype == torch.int32 else offset.int()
    noff = off if self_query else (new_offset if new_offset.dtype == torch.int32 else new_offset.int())
    off, noff = off.contiguous(), noff.contiguous()
    m, n, b = new_xyz.shape[0], xyz.shape[0], off.shape[0]
    idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
    dist2 = torch.empty((m, nsample), dtype=torch.float32, device=xyz.device)
    L = _lib.lib()
    if grid is not None and nsample <= 32 and n > 0 and m > 0:
        # sized for the largest query of a level: the interpolation table (m = the finer level) comes first in a scene
        need = L.knn_query_hip_workspace_bytes(max(m, n), n, b)
        mode, slot, ws = grid.plan(xyz, off, m, n, b, need, _lib.stream_ptr())
        rc = L.knn_query_grid_hip_launcher(m, nsample, xyz.data_ptr(), new_xyz.data_ptr(), off.data_ptr(), noff.data_ptr(),
                                           idx.data_ptr(), dist2.data_ptr(), n, b, int(pad_with_start), mode, slot,
                                           ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "knn_query_grid_hip_launcher")
        return idx, dist2
    ws = _lib.workspace(L.knn_query_hip_workspace_bytes(m, n, b), xyz.device)
    rc = L.knn_query_hip_launcher(m, nsample, xyz.data_ptr(), new_xyz.data_ptr(), off.data_ptr(), noff.data_ptr(),
                                  idx.data_ptr(), dist2.data_ptr(), n, b, int(pad_with_start), ws.data_ptr(),
                                  ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "knn_query_hip_launcher")
    return idx, dist2


class KNNQuery(Function):
    @staticmethod
    def forward(ctx, nsample, xyz, offset, new_xyz=None, new_offset=None):
        """
        input: coords: (n, 3), new_xyz: (m, 3), offset: (b), new_offset: (b)
        output: idx: (m, nsample) -1 is placeholder, dist: (m, nsample)
        """
        idx, dist2 = knn_query_dist2(nsample, xyz, offset, new_xyz, new_offset)
        ctx.mark_non_differentiable(idx)
        return idx, torch.sqrt(dist2)

    @staticmethod
    def backward(ctx, *grads):
        return None, None, None, None, None


knn_query = KNNQuery.apply


def ball_query(*args, **kwargs):
    raise NotImplementedError("ball_query is outside the PT-v2m2 hot path (SURVEY.md section 2a); not built")


def random_ball_query(*args, **kwargs):
    raise NotImplementedError("random_ball_query is outside the PT-v2m2 hot path (SURVEY.md section 2a); not built")
