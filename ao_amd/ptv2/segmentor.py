"""Thin caller-side pieces the bench/tests need around the backbone:
the segmentor wrapper (pointcept/models/default.py:232-251) and the two reference configs."""
import os

import torch
import torch.nn as nn

from .model import PointTransformerV2

S3DIS_BACKBONE = dict(  # configs/s3dis/semseg-pt-v2m2-0-base.py:12-36
    in_channels=6, num_classes=13, patch_embed_depth=2, patch_embed_channels=48, patch_embed_groups=6,
    patch_embed_neighbours=16, enc_depths=(2, 6, 2), enc_channels=(96, 192, 384), enc_groups=(12, 24, 48),
    enc_neighbours=(16, 16, 16), dec_depths=(1, 1, 1), dec_channels=(48, 96, 192), dec_groups=(6, 12, 24),
    dec_neighbours=(16, 16, 16), grid_sizes=(0.1, 0.2, 0.4), attn_qkv_bias=True, pe_multiplier=False, pe_bias=True,
    attn_drop_rate=0.0, drop_path_rate=0.3, enable_checkpoint=False, unpool_backend="interp")

SCANNET_BACKBONE = dict(  # configs/scannet/semseg-pt-v2m2-0-base.py:10-37
    in_channels=9, num_classes=20, patch_embed_depth=1, patch_embed_channels=48, patch_embed_groups=6,
    patch_embed_neighbours=8, enc_depths=(2, 2, 6, 2), enc_channels=(96, 192, 384, 512), enc_groups=(12, 24, 48, 64),
    enc_neighbours=(16, 16, 16, 16), dec_depths=(1, 1, 1, 1), dec_channels=(48, 96, 192, 384),
    dec_groups=(6, 12, 24, 48), dec_neighbours=(16, 16, 16, 16), grid_sizes=(0.06, 0.15, 0.375, 0.9375),
    attn_qkv_bias=True, pe_multiplier=False, pe_bias=True, attn_drop_rate=0.0, drop_path_rate=0.3,
    enable_checkpoint=False, unpool_backend="map")


class _CrossEntropy(torch.autograd.Function):
    """Mean softmax cross-entropy with ignore_index on ao_amd/csrc/loss.hip (fp32 CUDA logits (N,C), int64 labels)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, logits, label, ignore_index):
        from .. import _lib

        logits, label = logits.contiguous(), label.contiguous()
        n, c = logits.shape
        L = _lib.lib()
        dev = logits.device
        lse = torch.empty(n, dtype=torch.float32, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)  # loss, labelled count, out-of-range labels
        ws = _lib.workspace(L.cross_entropy_workspace_bytes(n), dev)
        rc = L.cross_entropy_forward_hip_launcher(n, c, logits.data_ptr(), label.data_ptr(), int(ignore_index), lse.data_ptr(),
                                                  out.data_ptr(), out.data_ptr() + 4, out.data_ptr() + 8, ws.data_ptr(),
                                                  ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "cross_entropy_forward_hip_launcher")
        if os.environ.get("AO_AMD_CHECK_LABELS") == "1" and float(out[2]) > 0:  # costs a synchronisation: debugging aid
            raise ValueError("cross_entropy: %d labels are neither ignore_index=%d nor in [0, %d)"
                             % (int(out[2]), int(ignore_index), c))
        ctx.save_for_backward(logits, label, lse, out)
        ctx.ignore_index = int(ignore_index)
        return out[0]

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        from .. import _lib

        logits, label, lse, out = ctx.saved_tensors
        n, c = logits.shape
        g = g.contiguous().float()
        gl = torch.empty_like(logits)
        rc = _lib.lib().cross_entropy_backward_hip_launcher(n, c, logits.data_ptr(), label.data_ptr(), ctx.ignore_index,
                                                            lse.data_ptr(), g.data_ptr(), out.data_ptr() + 4, gl.data_ptr(),
                                                            _lib.stream_ptr())
        _lib.check(rc, "cross_entropy_backward_hip_launcher")
        return gl, None, None


def cross_entropy(logits, label, ignore_index=-1):
    """F.cross_entropy(logits, label, ignore_index=...) (mean reduction) on the HIP kernel when it applies.

    A label that is neither `ignore_index` nor in [0, C) -- torch raises a device-side assert for it (e.g. a dataset
    that marks "unlabelled" with 255 while the loss ignores -1) -- makes the loss NaN here: no synchronisation is
    spent on the check, and the run fails at the first place the loss is looked at instead of silently training on
    fewer points.  AO_AMD_CHECK_LABELS=1 raises a ValueError at the call instead (one host synchronisation per step).
    With no labelled point at all the loss is NaN and the gradient zero, as torch."""
    if logits.is_cuda and logits.dim() == 2 and label.dtype == torch.int64 and logits.shape[0] > 0 and logits.shape[1] <= 1024:
        return _CrossEntropy.apply(logits, label, ignore_index)
    return torch.nn.functional.cross_entropy(logits, label, ignore_index=ignore_index)


def _parse_criteria(criteria, ignore_index):
    """The reference builds `criteria` from a list of loss configs and sums them (pointcept/models/losses/builder.py:
    13-29).  On this path every config uses ONE CrossEntropyLoss (configs/s3dis/semseg-pt-v2m2-0-base.py:37,
    semseg-pt-v2m2-0-sam-final.py:37); that is what is implemented: a list of CrossEntropyLoss entries, each with its
    loss_weight / ignore_index (losses/misc.py:14-39)."""
    if criteria is None:
        return [(1.0, ignore_index)]
    out = []
    for c in criteria:
        c = dict(c)
        kind = c.pop("type", "CrossEntropyLoss")
        extra = set(c) - {"loss_weight", "ignore_index"}
        if kind != "CrossEntropyLoss" or extra:
            raise NotImplementedError("criteria %r with %s: only CrossEntropyLoss(loss_weight, ignore_index) is on the "
                                      "PT-v2m2 path" % (kind, sorted(extra)))
        out.append((float(c.get("loss_weight", 1.0)), int(c.get("ignore_index", -1))))
    return out


class DefaultSegmentor(nn.Module):
    """backbone + CrossEntropyLoss(ignore_index=-1); constructor and return convention of the reference
    (pointcept/models/default.py:232-251): `DefaultSegmentor(backbone=dict(type="PT-v2m2", ...), criteria=[dict(
    type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1)])`."""

    @property
    def _ddp_params_and_buffers_to_ignore(self):
        """DistributedDataParallel wraps the segmentor (engines/train_sam_pp2s.py:207-213): see PointTransformerV2's property"""
        from .model import parallel_ddp_ignore

        return parallel_ddp_ignore(self, "backbone.") if isinstance(self.backbone, PointTransformerV2) else []

    def __init__(self, backbone=None, criteria=None, ignore_index=-1):
        super().__init__()
        self.backbone = backbone if isinstance(backbone, nn.Module) else PointTransformerV2(
            **{k: v for k, v in dict(backbone).items() if k != "type"})
        self._criteria = _parse_criteria(criteria, ignore_index)
        self.ignore_index = self._criteria[0][1]
        self.criteria = nn.CrossEntropyLoss(ignore_index=self.ignore_index)  # kept for state / introspection parity

    def loss(self, seg_logits, segment):
        total = None
        for weight, ignore in self._criteria:
            term = cross_entropy(seg_logits, segment, ignore)
            term = term if weight == 1.0 else term * weight
            total = term if total is None else total + term
        return total

    def forward(self, input_dict):
        seg_logits = self.backbone(input_dict)
        if self.training:
            return dict(loss=self.loss(seg_logits, input_dict["segment"]))
        if "segment" in input_dict:
            return dict(loss=self.loss(seg_logits, input_dict["segment"]), seg_logits=seg_logits)
        return dict(seg_logits=seg_logits)


class DefaultSegmentorSAM_Image(DefaultSegmentor):
    """REAL's segmentor (pointcept/models/default.py:15-76): in training it returns `(dict(loss=loss), seg_dict)` with
    `seg_dict[scene_key] = (seg_logits of that scene (n, C), original point ids of those rows (n,))`, where
    `scene_key = scene_id.replace("/", "_")[:-4]` (:53) and the ids come from `input_dict["instance"]` (:31); eval /
    test returns are DefaultSegmentor's.  The trainer copies seg_dict into its basket every step
    (engines/train_sam_real.py:229-234; here: ao_amd/ptv2/basket.LogitBasket.put, asynchronous).

    The reference slices with device-tensor bounds (:39-40) and calls `.unique()` per scene (:46-48, result unused):
    at least three host synchronisations per scene before the backward is issued.  Here the bounds come from
    `input_dict["offset_host"]` (a python list the collate function already has, ao_amd/ptv2/transform.collate) when
    present -- no synchronisation at all -- else from ONE `offset.tolist()`."""

    def __init__(self, backbone=None, criteria=None, ignore_index=-1):
        super().__init__(backbone, criteria, ignore_index)
        self.count = 0  # the reference trainer resets this attribute every epoch (train_sam_real.py:259)

    @staticmethod
    def scene_key(scene_id):
        return scene_id.replace("/", "_")[:-4]

    def forward(self, input_dict):
        seg_logits = self.backbone(input_dict)
        if self.training:
            seg_dict = {}
            with torch.no_grad():
                bounds = input_dict.get("offset_host")
                if bounds is None:
                    bounds = input_dict["offset"].tolist()
                original_idx, detached, start = input_dict["instance"], seg_logits.detach(), 0
                for scene_id, end in zip(input_dict["scene_id"], bounds):
                    end = int(end)
                    seg_dict[self.scene_key(scene_id)] = (detached[start:end], original_idx[start:end])
                    start = end
            return dict(loss=self.loss(seg_logits, input_dict["segment"])), seg_dict
        if "segment" in input_dict:
            return dict(loss=self.loss(seg_logits, input_dict["segment"]), seg_logits=seg_logits)
        return dict(seg_logits=seg_logits)
