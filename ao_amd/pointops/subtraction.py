"""Mirrors libs/pointops/functions/subtraction.py:7-38."""
import torch
from torch.autograd import Function

from .. import _lib


class Subtraction(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, input1, input2, idx):
        """
        input: input1: (n, c), input2: (n, c), idx: (n, nsample)
        output:  (n, nsample, c)
        """
        _lib.require_cuda(input1, input2, idx)
        assert input1.is_contiguous() and input2.is_contiguous()
        idx = idx.contiguous()
        n, c = input1.shape
        nsample = idx.shape[-1]
        output = torch.empty((n, nsample, c), dtype=torch.float32, device=input1.device)
        rc = _lib.lib().subtraction_forward_hip_launcher(n, nsample, c, input1.data_ptr(), input2.data_ptr(),
                                                         idx.data_ptr(), output.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "subtraction_forward_hip_launcher")
        ctx.save_for_backward(idx)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        (idx,) = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, nsample, c = grad_output.shape
        grad_input1 = torch.zeros((n, c), dtype=torch.float32, device=grad_output.device)
        grad_input2 = torch.zeros((n, c), dtype=torch.float32, device=grad_output.device)
        rc = _lib.lib().subtraction_backward_hip_launcher(n, nsample, c, idx.data_ptr(), grad_output.data_ptr(),
                                                          grad_input1.data_ptr(), grad_input2.data_ptr(),
                                                          _lib.stream_ptr())
        _lib.check(rc, "subtraction_backward_hip_launcher")
        return grad_input1, grad_input2, None


subtraction = Subtraction.apply
