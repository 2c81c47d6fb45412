"""`from pointops2 import pointops` / `import pointops2.pointops` compatible module."""
import torch

from ..pointops.aggregation import aggregation  # noqa: F401
from ..pointops.grouping import grouping2 as grouping  # noqa: F401  (pointops2 `grouping` is the CUDA op, :58-89)
from ..pointops.interpolation import interpolation_index_weight, _InterpolateRows
from ..pointops.query import knn_query_dist2
from ..pointops.sampling import farthest_point_sampling as furthestsampling  # noqa: F401
from ..pointops.subtraction import subtraction  # noqa: F401


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """pointops2 argument order (:36-55); placeholder index is the segment start (knnquery_cuda_kernel.cu:90)."""
    if new_xyz is None:
        new_xyz, new_offset = xyz, offset
    idx, dist2 = knn_query_dist2(nsample, xyz, offset, new_xyz, new_offset, pad_with_start=True)
    return idx, torch.sqrt(dist2)


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True, return_indx=False):
    """:963-1001 -- plain gather (no -1 masking), relative xyz prepended."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        idx, _ = knnquery(nsample, xyz, new_xyz, offset, new_offset)
    grouped_feat = grouping(feat, idx)
    out = grouped_feat
    if use_xyz:
        grouped_xyz = grouping(xyz, idx) - new_xyz.unsqueeze(1)
        out = torch.cat((grouped_xyz, grouped_feat), -1)
    return (out, idx) if return_indx else out


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """:1112-1127"""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    idx, weight = interpolation_index_weight(xyz, new_xyz, offset, new_offset, k)
    return _InterpolateRows.apply(feat, idx, weight)


interpolation2 = interpolation
