"""Validation arithmetic of the segmentation path: the k = 1 label transfer from the
grid-sampled cloud back to the original points, the per-class intersection / union / target counts and their
reduction over ranks -- pointcept/engines/hooks/evaluator.py:111-175 and pointcept/utils/misc.py:58-70.

One HIP kernel (ao_amd/csrc/dataops.hip: seg_confusion_kernel) reads the prediction through the nearest-neighbour
table and builds the three histograms in int64: exact counts (the reference's torch.histc returns floats, exact up to
2^24 per class), no intermediate (N,) tensors, and the reference's in-place masking of `output` is not performed on
the caller's tensor.
"""
import numpy as np
import torch

from .. import _lib, pointops


def confusion_counts(pred, target, k, ignore_index=-1, nn_idx=None):
    """(3, k) int64 device tensor: intersection, output area, target area.  pred: (M,) integer class ids; target: (N,);
    nn_idx: (N,) int32 row of pred for every target element (None: N == M, identity)."""
    if not (pred.is_cuda and target.is_cuda):
        raise RuntimeError("ao_amd.ptv2.evaluate works on CUDA tensors (no CPU fallback)")
    pred = pred.reshape(-1).long().contiguous()
    target = target.reshape(-1).long().contiguous()
    if nn_idx is None:
        assert pred.shape == target.shape
    else:
        nn_idx = nn_idx.reshape(-1).int().contiguous()
        assert nn_idx.shape == target.shape
    hist = torch.empty((3, k), dtype=torch.int64, device=target.device)
    rc = _lib.lib().seg_confusion_hip_launcher(target.numel(), int(k), int(ignore_index), pred.data_ptr(), pred.numel(),
                                               nn_idx.data_ptr() if nn_idx is not None else None, target.data_ptr(),
                                               hist.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "seg_confusion_hip_launcher")
    return hist


def intersection_and_union_gpu(output, target, k, ignore_index=-1):
    """misc.py:58-70: (area_intersection, area_union, area_target), each (k,)."""
    assert output.dim() in [1, 2, 3]
    assert output.shape == target.shape
    h = confusion_counts(output, target, k, ignore_index)
    return h[0], h[1] + h[2] - h[0], h[2]


def evaluate_batch(output_dict, input_dict, num_classes, ignore_index=-1, group=None):
    """evaluator.py:119-141 for one validation batch: (intersection, union, target) as int64 numpy arrays, summed over
    the ranks of `group` when torch.distributed is initialised."""
    pred = output_dict["seg_logits"].max(1)[1]
    segment, nn_idx = input_dict["segment"], None
    if "origin_coord" in input_dict.keys():
        nn_idx, _ = pointops.knn_query(1, input_dict["coord"].float(), input_dict["offset"].int(),
                                       input_dict["origin_coord"].float(), input_dict["origin_offset"].int())
        segment = input_dict["origin_segment"]
    h = confusion_counts(pred, segment, num_classes, ignore_index, nn_idx)
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1:
        torch.distributed.all_reduce(h, group=group)
    h = h.cpu().numpy()
    return h[0], h[1] + h[2] - h[0], h[2]


def summarize(intersection, union, target):
    """evaluator.py:161-171: dict(mIoU, mAcc, allAcc, iou_class, acc_class) from the totals over the validation set."""
    intersection, union, target = (np.asarray(a, dtype=np.float64) for a in (intersection, union, target))
    iou_class = intersection / (union + 1e-10)
    acc_class = intersection / (target + 1e-10)
    return dict(mIoU=float(np.mean(iou_class)), mAcc=float(np.mean(acc_class)),
                allAcc=float(sum(intersection) / (sum(target) + 1e-10)), iou_class=iou_class, acc_class=acc_class)
