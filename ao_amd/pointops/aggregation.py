"""Mirrors libs/pointops/functions/aggregation.py:7-57."""
import torch
from torch.autograd import Function

from .. import _lib


class Aggregation(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, input, position, weight, idx):
        """
        input: input: (n, c), position: (n, nsample, c), weight : (n, nsample, c'), idx: (n, nsample)
        output: (n, c)
        """
        _lib.require_cuda(input, position, weight, idx)
        assert input.is_contiguous() and position.is_contiguous() and weight.is_contiguous()
        idx = idx.contiguous()
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        output = torch.zeros((n, c), dtype=torch.float32, device=input.device)
        rc = _lib.lib().aggregation_forward_hip_launcher(n, nsample, c, w_c, input.data_ptr(), position.data_ptr(),
                                                         weight.data_ptr(), idx.data_ptr(), output.data_ptr(),
                                                         _lib.stream_ptr())
        _lib.check(rc, "aggregation_forward_hip_launcher")
        ctx.save_for_backward(input, position, weight, idx)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        input, position, weight, idx = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        dev = grad_output.device
        grad_input = torch.zeros((n, c), dtype=torch.float32, device=dev)
        grad_position = torch.zeros((n, nsample, c), dtype=torch.float32, device=dev)
        grad_weight = torch.zeros((n, nsample, w_c), dtype=torch.float32, device=dev)
        rc = _lib.lib().aggregation_backward_hip_launcher(
            n, nsample, c, w_c, input.data_ptr(), position.data_ptr(), weight.data_ptr(), idx.data_ptr(),
            grad_output.data_ptr(), grad_input.data_ptr(), grad_position.data_ptr(), grad_weight.data_ptr(),
            _lib.stream_ptr())
        _lib.check(rc, "aggregation_backward_hip_launcher")
        return grad_input, grad_position, grad_weight, None


aggregation = Aggregation.apply
