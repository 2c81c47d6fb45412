"""Thin caller-side pieces the bench/tests need around the backbone:
the segmentor wrapper (pointcept/models/default.py:232-251) and the two reference configs."""
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


class DefaultSegmentor(nn.Module):
    """backbone + CrossEntropyLoss(ignore_index=-1); same return convention as the reference."""

    def __init__(self, backbone=None, ignore_index=-1):
        super().__init__()
        self.backbone = backbone if isinstance(backbone, nn.Module) else PointTransformerV2(
            **{k: v for k, v in dict(backbone).items() if k != "type"})
        self.criteria = nn.CrossEntropyLoss(ignore_index=ignore_index)

    def forward(self, input_dict):
        seg_logits = self.backbone(input_dict)
        if self.training:
            return dict(loss=self.criteria(seg_logits, input_dict["segment"]))
        if "segment" in input_dict:
            return dict(loss=self.criteria(seg_logits, input_dict["segment"]), seg_logits=seg_logits)
        return dict(seg_logits=seg_logits)
