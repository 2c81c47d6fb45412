"""tests/golden/make_golden_host.py -- golden fixtures for the host-side rows of SURVEY.md §8f (GridSample,
SphereCrop, validation histograms, LR schedules), produced by RUNNING the reference's own Python in the build
container.  Nothing here is read at test time except the host_*.npz files it writes.

Executed from /root/reference, unmodified, on CPU:
  * pointcept/datasets/transform.py   GridSample (train + test mode, fnv + ravel hash), SphereCrop (center + random)
  * pointcept/utils/misc.py           intersection_and_union, intersection_and_union_gpu (on CPU float tensors)
  * pointcept/utils/scheduler.py      every registered scheduler, driven like engines/train.py:184-196
(torch 2.10 no longer accepts the `verbose=` argument scheduler.py forwards: accept_verbose_kwarg() drops it.)
The `pointcept` / `pointcept.utils` / `pointcept.datasets` package objects are empty namespace stubs (their
__init__.py would import the whole training stack); the modules above are loaded from their own files.

numpy note: GridSample is built with grid_size=np.float32(g): under this container's numpy 2.2 that makes
`coord / np.array(grid_size)` (transform.py:794) the fp32 division it was under the numpy 1.x of the reference's time.

usage:  python tests/golden/make_golden_host.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from tests import synth  # noqa: E402


def namespace_stubs():
    for name, sub in (("pointcept", "pointcept"), ("pointcept.utils", "pointcept/utils"),
                      ("pointcept.datasets", "pointcept/datasets")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m


def accept_verbose_kwarg():
    """scheduler.py passes verbose= to torch.optim.lr_scheduler.* (torch 1.12); torch 2.10 removed that argument.
    Drop it at the torch boundary so that the reference classes construct; nothing else is altered."""
    import torch.optim.lr_scheduler as L

    for name in ("MultiStepLR", "LambdaLR", "CosineAnnealingLR", "OneCycleLR"):
        base = getattr(L, name)

        def make(base):
            class Shim(base):
                def __init__(self, *a, verbose=False, **k):
                    super().__init__(*a, **k)

            Shim.__name__ = base.__name__
            return Shim

        setattr(L, name, make(base))


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrays.items()})


def dense_cloud(n, seed):
    """A room cloud with several points per 4 cm voxel (the raw-scan regime GridSample is meant for)."""
    rng = np.random.default_rng(seed)
    base = synth.room_cloud(n // 3, seed=seed)
    pts = np.concatenate([base + rng.normal(0, 0.012, base.shape).astype(np.float32) for _ in range(3)])
    pts = pts[rng.permutation(pts.shape[0])]
    return np.ascontiguousarray(pts - np.float32(1.7), dtype=np.float32)  # negative coordinates too (floor, not trunc)


def gen_gridsample(T):
    out = {}
    coord = dense_cloud(9000, 5)
    out["coord"] = coord
    for tag, hash_type, g in (("fnv", "fnv", 0.04), ("ravel", "ravel", 0.05)):
        gs = T.GridSample(grid_size=np.float32(g), hash_type=hash_type, mode="train", keys=("coord", "index"),
                          return_discrete_coord=True, return_min_coord=True)
        np.random.seed(11)
        d = gs(dict(coord=coord.copy(), index=np.arange(coord.shape[0])))
        out[tag + "_grid"] = np.float32(g)
        out[tag + "_train_index"] = d["index"]
        out[tag + "_train_cell"] = d["discrete_coord"]
        out[tag + "_min_coord"] = d["min_coord"]
        gt = T.GridSample(grid_size=np.float32(g), hash_type=hash_type, mode="test", keys=("coord", "index"))
        parts = gt(dict(coord=coord.copy(), index=np.arange(coord.shape[0])))
        out[tag + "_test_index"] = np.concatenate([p["index"] for p in parts])
        out[tag + "_test_sizes"] = np.asarray([p["index"].shape[0] for p in parts])
    # transform.py:807-815: the data-efficient branch (labelled points always kept).  It spells the mask dtype `np.bool`, an
    # alias numpy removed in 1.24: restored for this call only, nothing else of the reference is touched
    gs = T.GridSample(grid_size=np.float32(0.04), hash_type="fnv", mode="train", keys=("coord", "index", "segment"))
    labelled = np.random.default_rng(23).choice(coord.shape[0], size=200, replace=False)
    had = hasattr(np, "bool")
    if not had:
        np.bool = bool
    np.random.seed(13)
    d = gs(dict(coord=coord.copy(), index=np.arange(coord.shape[0]), segment=np.arange(coord.shape[0]) % 7,
                sampled_index=labelled.copy()))
    if not had:
        del np.bool
    out["sampled_in"] = labelled
    out["sampled_train_index"] = d["index"]
    out["sampled_out"] = d["sampled_index"]
    cells = np.random.default_rng(3).integers(0, 400, size=(500, 3))
    out["hash_cells"] = cells
    out["hash_fnv"] = T.GridSample.fnv_hash_vec(cells)
    out["hash_ravel"] = T.GridSample.ravel_hash_vec(cells)
    save("host_gridsample.npz", **out)


def gen_spherecrop(T):
    coord = synth.room_cloud(6000, seed=9)
    out = {"coord": coord}
    sc = T.SphereCrop(point_max=2500, mode="center")
    d = sc(dict(coord=coord.copy(), segment=np.arange(coord.shape[0])))
    out["center_index"] = d["segment"]
    sc = T.SphereCrop(point_max=1000, mode="random")
    np.random.seed(21)
    d = sc(dict(coord=coord.copy(), segment=np.arange(coord.shape[0])))
    np.random.seed(21)
    out["random_center"] = np.random.randint(coord.shape[0])
    out["random_index"] = d["segment"]
    d = T.SphereCrop(point_max=10000, mode="random")(dict(coord=coord.copy(), segment=np.arange(coord.shape[0])))
    out["nocrop_index"] = d["segment"]
    # mode="all" (test-time tiling, transform.py:914-968): every crop's members and weights, and the priorities it drew
    np.random.seed(31)
    out["all_priority"] = np.random.rand(coord.shape[0]) * 1e-3
    np.random.seed(31)
    parts = T.SphereCrop(point_max=2000, mode="all")(dict(coord=coord.copy(), color=coord.copy() * 2))
    out["all_sizes"] = np.asarray([p["index"].shape[0] for p in parts])
    out["all_index"] = np.concatenate([p["index"] for p in parts])
    out["all_weight"] = np.concatenate([p["weight"] for p in parts])
    out["all_color0"] = parts[0]["color"]
    save("host_spherecrop.npz", **out)


def gen_iou(M):
    rng = np.random.default_rng(17)
    out = {}
    for tag, n, k in (("s3dis", 20000, 13), ("scannet", 7001, 20)):
        target = rng.integers(-1, k, size=n)
        pred = np.where(rng.random(n) < 0.7, np.maximum(target, 0), rng.integers(0, k, size=n))
        i, u, t = M.intersection_and_union(pred.copy(), target.copy(), k, -1)
        gi, gu, gt = M.intersection_and_union_gpu(torch.from_numpy(pred.copy()).float(), torch.from_numpy(target.copy()).float(), k, -1)
        assert np.array_equal(gi.numpy(), i) and np.array_equal(gu.numpy(), u) and np.array_equal(gt.numpy(), t)
        out.update({tag + "_pred": pred, tag + "_target": target, tag + "_k": k, tag + "_intersection": i,
                    tag + "_union": u, tag + "_target_area": t})
    save("host_iou.npz", **out)


def gen_schedules(S):
    out = {}
    cases = {
        "s3dis": ("MultiStepLR", 0.006, dict(milestones=[0.09, 0.2], gamma=0.1)),
        "s3dis_late": ("MultiStepLR", 0.006, dict(milestones=[0.6, 0.8], gamma=0.1)),
        "warmup": ("MultiStepWithWarmupLR", 0.01, dict(milestones=[0.5, 0.75], gamma=0.1, warmup_rate=0.05, warmup_scale=1e-6)),
        "poly": ("PolyLR", 0.02, dict(power=0.9)),
        "exp": ("ExpLR", 0.02, dict(gamma=0.9)),
        "cosine": ("CosineAnnealingLR", 0.02, dict(eta_min=1e-5)),
        "scannet": ("OneCycleLR", 0.005, dict(max_lr=0.005, pct_start=0.05, anneal_strategy="cos", div_factor=10.0,
                                               final_div_factor=1000.0)),
        "onecycle_linear": ("OneCycleLR", 0.01, dict(max_lr=0.01, pct_start=0.3, anneal_strategy="linear")),
    }
    for total in (200, 333):
        for tag, (kind, lr, kw) in cases.items():
            w = torch.nn.Parameter(torch.zeros(3))
            opt = torch.optim.AdamW([w], lr=lr, weight_decay=0.05)
            sched = getattr(S, kind)(opt, total_steps=total, **kw)
            lrs, moms = [], []
            for _ in range(total):
                lrs.append(opt.param_groups[0]["lr"])
                moms.append(opt.param_groups[0]["betas"][0])
                w.grad = torch.ones(3)
                opt.step()
                sched.step()
            out["%s_%d_lr" % (tag, total)] = np.asarray(lrs, dtype=np.float64)
            out["%s_%d_beta1" % (tag, total)] = np.asarray(moms, dtype=np.float64)
    save("host_schedules.npz", **out)


def main():
    assert os.path.isdir(REF), "run in the build container: needs /root/reference"
    namespace_stubs()
    T = importlib.import_module("pointcept.datasets.transform")
    M = importlib.import_module("pointcept.utils.misc")
    accept_verbose_kwarg()
    S = importlib.import_module("pointcept.utils.scheduler")
    assert T.__file__.startswith(REF) and M.__file__.startswith(REF) and S.__file__.startswith(REF)
    gen_gridsample(T)
    gen_spherecrop(T)
    gen_iou(M)
    gen_schedules(S)


if __name__ == "__main__":
    main()
