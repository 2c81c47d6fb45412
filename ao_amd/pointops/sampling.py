"""Farthest point sampling (ao_amd/csrc/fps.hip); mirrors libs/pointops/functions/sampling.py:7-27."""
import torch
from torch.autograd import Function

from .. import _lib


class FarthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, offset, new_offset):
        """
        input: coords: (n, 3), offset: (b), new_offset: (b)
        output: idx: (m)
        """
        _lib.require_cuda(xyz, offset, new_offset)
        assert xyz.is_contiguous() and xyz.dtype == torch.float32
        n, b = xyz.shape[0], offset.shape[0]
        # the reference syncs here too (python max() over device scalars, sampling.py:15-17);
        # n_max fixes the reference's block size and thereby its tie rule
        sizes = torch.diff(offset.long(), prepend=offset.new_zeros(1, dtype=torch.long))
        n_max, m_total = int(sizes.max().item()), int(new_offset[b - 1].item())
        idx = torch.zeros(m_total, dtype=torch.int32, device=xyz.device)
        tmp = torch.full((n,), 1e10, dtype=torch.float32, device=xyz.device)
        L = _lib.lib()
        ws = _lib.workspace(L.farthest_point_sampling_hip_workspace_bytes(b, n), xyz.device)
        # the multi-workgroup form exchanges its per-iteration arg-max between workgroups that must all be resident; its
        # spins are bounded and raise this flag instead of hanging (ao_amd/csrc/fps.hip: error_flag, behind the granule
        # slots).  FPS costs tens of milliseconds and the reference synchronises around it as well: read the flag back.
        at = 8 * 2 * 4 * 256 * b  # behind the granule slots: [cloud][parity][key, x, y, z][256] of 8 bytes
        flag = ws[at:at + 4].view(torch.int32)
        flag.zero_()
        rc = L.farthest_point_sampling_hip_launcher(
            b, n_max, xyz.data_ptr(), offset.int().contiguous().data_ptr(), new_offset.int().contiguous().data_ptr(),
            tmp.data_ptr(), idx.data_ptr(), n, m_total, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "farthest_point_sampling_hip_launcher")
        code = int(flag.item())
        if code != 0:
            raise RuntimeError("ao_amd: farthest_point_sampling: a workgroup gave up waiting for its peers (the cooperative "
                               "kernel needs all its workgroups resident; other work was holding the compute units)")
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad):
        return None, None, None


farthest_point_sampling = FarthestPointSampling.apply
