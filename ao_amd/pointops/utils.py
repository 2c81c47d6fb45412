"""Mirrors libs/pointops/functions/utils.py:5-119 (ball_query_and_group is out of scope)."""
import torch

from .grouping import grouping
from .query import knn_query


def knn_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, nsample=None,
                        with_xyz=False):
    if idx is None:
        assert nsample is not None
        idx, _ = knn_query(nsample, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def ball_query_and_group(*args, **kwargs):
    raise NotImplementedError("ball_query is outside the PT-v2m2 hot path (SURVEY.md section 2a); not built")


def query_and_group(nsample, xyz, new_xyz, feat, idx, offset, new_offset, dilation=0, with_feat=True,
                    with_xyz=True):
    """
    input: coords: (n, 3), new_xyz: (m, 3), color: (n, c), idx: (m, nsample), offset: (b), new_offset: (b)
    output: new_feat: (m, nsample, c+3), grouped_idx: (m, nsample)
    """
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    if idx is None:
        total = 1 + (nsample - 1) * (dilation + 1)
        idx_all, _ = knn_query(total, xyz, offset, new_xyz, new_offset)
        ends, new_ends = offset.tolist(), new_offset.tolist()
        starts, new_starts = [0] + ends[:-1], [0] + new_ends[:-1]
        parts = []
        for i in range(len(ends)):
            cnt = ends[i] - starts[i]
            soft = (cnt - 1) / (nsample - 1) - 1 if cnt < total else dilation  # utils.py:74-77
            cols = [int((soft + 1) * j) for j in range(nsample)]
            parts.append(idx_all[new_starts[i]:new_ends[i], cols])
        idx = torch.cat(parts, dim=0).contiguous()
    if not with_feat:
        return idx
    grouped = grouping(idx, feat, xyz, new_xyz, with_xyz)
    return grouped, idx


def offset2batch(offset):
    """Same result as the reference's python loop (utils.py:96-109) without per-cloud host syncs."""
    off = offset.long()
    counts = torch.diff(off, prepend=off.new_zeros(1))
    return torch.repeat_interleave(torch.arange(off.numel(), device=off.device), counts)


def batch2offset(batch):
    return torch.cumsum(batch.bincount(), dim=0).int()
