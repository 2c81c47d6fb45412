"""Neighbour gather.  Mirrors libs/pointops/functions/grouping.py:7-63.

`grouping2` is the reference's CUDA op (no -1 handling there; here -1 yields zeros).
`grouping` is the reference's pure-torch function (zero row for -1, masked relative xyz); it
runs on the same HIP gather kernel instead of torch.cat + fancy indexing.
"""
import torch
from torch.autograd import Function

from .. import _lib


class Grouping(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, input, idx):
        """
        input: input: (n, c), idx : (m, nsample)
        output: (m, nsample, c)
        """
        _lib.require_cuda(input, idx)
        assert input.is_contiguous() and idx.is_contiguous()
        assert input.dtype == torch.float32 and idx.dtype == torch.int32
        m, nsample, n, c = idx.shape[0], idx.shape[1], input.shape[0], input.shape[1]
        output = torch.empty((m, nsample, c), dtype=torch.float32, device=input.device)
        rc = _lib.lib().grouping_forward_hip_launcher(m, nsample, c, input.data_ptr(), idx.data_ptr(),
                                                      output.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "grouping_forward_hip_launcher")
        ctx.n = n
        ctx.save_for_backward(idx)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        (idx,) = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        m, nsample, c = grad_output.shape
        grad_input = torch.zeros((ctx.n, c), dtype=torch.float32, device=grad_output.device)
        rc = _lib.lib().grouping_backward_hip_launcher(m, nsample, c, grad_output.data_ptr(), idx.data_ptr(),
                                                       grad_input.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "grouping_backward_hip_launcher")
        return grad_input, None


grouping2 = Grouping.apply


def grouping(idx, feat, xyz, new_xyz=None, with_xyz=False):
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and feat.is_contiguous()
    idx = idx.contiguous()
    grouped_feat = Grouping.apply(feat, idx)  # (m, nsample, c), zeros where idx == -1
    if not with_xyz:
        return grouped_feat
    assert new_xyz.is_contiguous()
    mask = torch.sign(idx + 1).to(xyz.dtype).unsqueeze(-1)
    grouped_xyz = (Grouping.apply(xyz, idx) - new_xyz.unsqueeze(1)) * mask
    return torch.cat((grouped_xyz, grouped_feat), -1)
