"""Inverse-distance interpolation.  Mirrors libs/pointops/functions/interpolation.py:8-59."""
import torch
from torch.autograd import Function

from .. import _lib
from .query import knn_query, knn_query_dist2


def interpolation_index_weight(xyz, new_xyz, offset, new_offset, k=3, grid=None):
    """k-NN of every new_xyz row among xyz + normalised inverse-distance weights (:13-16).
    Index -1 (coarse segment shorter than k) wraps to the last row as torch indexing does (:21)."""
    if k <= 8:  # one launch (interpolation_weights_hip_launcher: the same arithmetic in the same order) instead of nine
        idx, dist2 = knn_query_dist2(k, xyz, offset, new_xyz, new_offset, grid=grid)  # grid: the source points' KnnGrid
        weight = torch.empty_like(dist2)
        rc = _lib.lib().interpolation_weights_hip_launcher(idx.shape[0], k, xyz.shape[0], dist2.data_ptr(), idx.data_ptr(),
                                                           weight.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "interpolation_weights_hip_launcher")
        return idx, weight
    idx, dist = knn_query(k, xyz, offset, new_xyz, new_offset)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=1, keepdim=True)
    weight = (dist_recip / norm).contiguous()
    idx = torch.where(idx < 0, idx + xyz.shape[0], idx).contiguous()
    return idx, weight


class _InterpolateRows(Function):
    """output[n,:] = sum_i input[idx[n,i],:] * weight[n,i] with the scatter-add backward."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, input, idx, weight):
        _lib.require_cuda(input, idx, weight)
        input = input.contiguous()
        n, k = idx.shape
        m, c = input.shape
        output = torch.zeros((n, c), dtype=torch.float32, device=input.device)
        rc = _lib.lib().interpolation_forward_hip_launcher(n, c, k, input.data_ptr(), idx.data_ptr(),
                                                           weight.data_ptr(), output.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "interpolation_forward_hip_launcher")
        ctx.m = m
        ctx.save_for_backward(idx, weight)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        idx, weight = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, c = grad_output.shape
        inv = getattr(idx, "_ao_inverse", None)  # built with the geometry (ao_amd/ptv2/geometry.py): gather, no atomics
        if inv is not None and inv[0] == idx._version and ctx.m <= n:
            grad_input = torch.empty((ctx.m, c), dtype=torch.float32, device=grad_output.device)
            rc = _lib.lib().interpolation_backward_gather_hip_launcher(ctx.m, c, idx.shape[1], grad_output.data_ptr(),
                                                                       inv[1].data_ptr(), inv[2].data_ptr(), weight.data_ptr(),
                                                                       grad_input.data_ptr(), _lib.stream_ptr())
            _lib.check(rc, "interpolation_backward_gather_hip_launcher")
            return grad_input, None, None
        grad_input = torch.zeros((ctx.m, c), dtype=torch.float32, device=grad_output.device)
        rc = _lib.lib().interpolation_backward_hip_launcher(n, c, idx.shape[1], grad_output.data_ptr(), idx.data_ptr(),
                                                            weight.data_ptr(), grad_input.data_ptr(),
                                                            _lib.stream_ptr())
        _lib.check(rc, "interpolation_backward_hip_launcher")
        return grad_input, None, None


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """
    input: coords: (m, 3), new_xyz: (n, 3), color: (m, c), offset: (b), new_offset: (b)
    output: (n, c)
    """
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    idx, weight = interpolation_index_weight(xyz, new_xyz, offset, new_offset, k)
    return _InterpolateRows.apply(feat, idx, weight)


def interpolation2(xyz, new_xyz, input, offset, new_offset, k=3):
    """The reference's autograd.Function spelling (:25-59); same computation."""
    return interpolation(xyz, new_xyz, input, offset, new_offset, k)
