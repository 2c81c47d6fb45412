"""Edge-list grouped attention steps.  Mirrors libs/pointops/functions/attention.py:12-120."""
import torch
from torch.autograd import Function

from .. import _lib


class AttentionRelationStep(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, query, key, weight, index_target, index_refer):
        """
        input - query: (n, g, c), key: (n, g, c), weight: (c)  1_c for scatter attention,
                index_target: (m), index_refer: (m)
        output - relation: (M, g)
        """
        _lib.require_cuda(query, key, weight, index_target, index_refer)
        assert query.is_contiguous() and key.is_contiguous() and weight.is_contiguous()
        assert index_target.is_contiguous() and index_refer.is_contiguous()
        assert index_target.shape[0] == index_refer.shape[0]
        _, g, c = query.shape
        m = index_target.shape[0]
        tgt, ref = index_target.int(), index_refer.int()
        output = torch.zeros((m, g), dtype=torch.float32, device=query.device)
        rc = _lib.lib().attention_relation_step_forward_hip_launcher(
            m, g, c, query.data_ptr(), key.data_ptr(), weight.data_ptr(), tgt.data_ptr(), ref.data_ptr(),
            output.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "attention_relation_step_forward_hip_launcher")
        ctx.save_for_backward(query, key, weight, tgt, ref)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        query, key, weight, tgt, ref = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, g, c = query.shape
        m = tgt.shape[0]
        grad_query = torch.zeros_like(query)
        grad_key = torch.zeros_like(key)
        grad_weight = torch.zeros_like(weight)
        rc = _lib.lib().attention_relation_step_backward_hip_launcher(
            m, g, c, query.data_ptr(), grad_query.data_ptr(), key.data_ptr(), grad_key.data_ptr(), weight.data_ptr(),
            grad_weight.data_ptr(), tgt.data_ptr(), ref.data_ptr(), grad_output.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "attention_relation_step_backward_hip_launcher")
        return grad_query, grad_key, None, None, None  # the reference also drops grad_weight (:63)


class AttentionFusionStep(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, weight, value, index_target, index_refer):
        """
        input - weight: (m, g), value: (n, g, c)
                index_target: (m), index_value: (m)
        output - output: (n, g, c)
        """
        _lib.require_cuda(weight, value, index_target, index_refer)
        assert weight.is_contiguous() and value.is_contiguous()
        assert index_target.is_contiguous() and index_refer.is_contiguous()
        assert index_target.shape[0] == index_refer.shape[0]
        n, g, c = value.shape
        m = index_refer.shape[0]
        tgt, ref = index_target.int(), index_refer.int()
        output = torch.zeros((n, g, c), dtype=torch.float32, device=value.device)
        rc = _lib.lib().attention_fusion_step_forward_hip_launcher(
            m, g, c, weight.data_ptr(), value.data_ptr(), tgt.data_ptr(), ref.data_ptr(), output.data_ptr(),
            _lib.stream_ptr())
        _lib.check(rc, "attention_fusion_step_forward_hip_launcher")
        ctx.save_for_backward(weight, value, tgt, ref)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        weight, value, tgt, ref = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, g, c = value.shape
        m = tgt.shape[0]
        grad_weight = torch.zeros_like(weight)
        grad_value = torch.zeros_like(value)
        rc = _lib.lib().attention_fusion_step_backward_hip_launcher(
            m, g, c, weight.data_ptr(), grad_weight.data_ptr(), value.data_ptr(), grad_value.data_ptr(),
            tgt.data_ptr(), ref.data_ptr(), grad_output.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "attention_fusion_step_backward_hip_launcher")
        return grad_weight, grad_value, None, None


attention_relation_step = AttentionRelationStep.apply
attention_fusion_step = AttentionFusionStep.apply
