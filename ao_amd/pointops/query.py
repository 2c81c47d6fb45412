"""k-NN query on the HIP grid kernel (ao_amd/csrc/knn.hip).

Mirrors libs/pointops/functions/query.py:7-24,111: same name, argument order, dtypes, -1 / 1e10
placeholders and the sqrt on the way out.  ball_query / random_ball_query are outside the
PT-v2m2 path and raise.
"""
import torch
from torch.autograd import Function

from .. import _lib


def knn_query_dist2(nsample, xyz, offset, new_xyz=None, new_offset=None, pad_with_start=False):
    """(idx int32 (m,k), squared distances fp32 (m,k)) -- the raw kernel outputs."""
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
