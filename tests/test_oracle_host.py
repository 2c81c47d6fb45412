"""CPU: oracle/host_ref.py (GridSample, SphereCrop, validation histograms, LR schedules) against the fixtures that
tests/golden/make_golden_host.py captured from the reference's own Python."""
import os

import numpy as np
import pytest

from oracle import host_ref as H
from tests.conftest import ROOT

G = os.path.join(ROOT, "tests", "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_voxel_hashes_match_reference():
    g = load("host_gridsample.npz")
    assert np.array_equal(H.fnv_hash_vec(g["hash_cells"]), g["hash_fnv"])
    assert np.array_equal(H.ravel_hash_vec(g["hash_cells"]), g["hash_ravel"])


@pytest.mark.parametrize("tag", ["fnv", "ravel"])
def test_grid_sample_train_matches_reference(tag):
    g = load("host_gridsample.npz")
    coord, grid = g["coord"], float(g[tag + "_grid"])
    idx_sort, uniq, count, cell, lo = H.grid_sample_sorted(coord, grid, tag)
    ref_idx = g[tag + "_train_index"]
    # what the reference determines: one point per voxel, voxels in ascending key order, their cells, the min corner
    assert ref_idx.shape[0] == uniq.shape[0]
    key = H.fnv_hash_vec(cell) if tag == "fnv" else H.ravel_hash_vec(cell)
    assert np.array_equal(key[ref_idx], uniq)
    assert np.array_equal(cell[ref_idx], g[tag + "_train_cell"])
    np.testing.assert_allclose(lo * np.float32(grid), g[tag + "_min_coord"][0], rtol=1e-6)
    # same draws (legacy numpy RandomState is reproducible): identical voxels, identical points where a voxel holds one
    np.random.seed(11)
    draws = np.random.randint(0, count.max(), count.size)
    ours = H.grid_sample_train(coord, grid, draws, tag)
    assert np.array_equal(key[ours], uniq)
    single = count == 1
    assert single.sum() > 100
    assert np.array_equal(ours[single], ref_idx[single])


@pytest.mark.parametrize("tag", ["fnv", "ravel"])
def test_grid_sample_test_mode_matches_reference(tag):
    g = load("host_gridsample.npz")
    coord, grid = g["coord"], float(g[tag + "_grid"])
    parts = H.grid_sample_test(coord, grid, tag)
    sizes = g[tag + "_test_sizes"]
    assert [p.shape[0] for p in parts] == list(sizes)
    ref = np.split(g[tag + "_test_index"], np.cumsum(sizes)[:-1])
    cell, _ = H.grid_cells(coord, grid)
    covered = np.zeros(coord.shape[0], bool)
    for a, b in zip(parts, ref):
        assert np.array_equal(cell[a], cell[b])  # same voxel in every slot of every part
        covered[a] = True
    assert covered.all()  # the parts cover the cloud, as the reference's do
    assert np.array_equal(np.unique(np.concatenate(parts)), np.unique(g[tag + "_test_index"]))


def test_sphere_crop_matches_reference():
    g = load("host_spherecrop.npz")
    coord = g["coord"]
    for ref_idx, centre, pmax in ((g["center_index"], coord.shape[0] // 2, 2500), (g["random_index"], int(g["random_center"]), 1000)):
        ours = H.sphere_crop(coord, pmax, centre)
        d2 = H.center_dist2(coord, coord[centre])
        assert np.array_equal(d2[ours], d2[ref_idx])  # same distance sequence (ties may swap equidistant points)
        assert np.array_equal(np.sort(ours), np.sort(ref_idx)) or d2[ours][-1] == np.sort(d2)[pmax]  # boundary tie only
    assert np.array_equal(H.sphere_crop(coord, 10000, 0), g["nocrop_index"])


@pytest.mark.parametrize("tag", ["s3dis", "scannet"])
def test_intersection_and_union_matches_reference(tag):
    g = load("host_iou.npz")
    i, u, t = H.intersection_and_union(g[tag + "_pred"], g[tag + "_target"], int(g[tag + "_k"]), -1)
    assert np.array_equal(i, g[tag + "_intersection"])
    assert np.array_equal(u, g[tag + "_union"])
    assert np.array_equal(t, g[tag + "_target_area"])


CASES = {
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


@pytest.mark.parametrize("total", [200, 333])
@pytest.mark.parametrize("tag", sorted(CASES))
def test_lr_curves_match_reference(tag, total):
    g = load("host_schedules.npz")
    kind, lr, kw = CASES[tag]
    got = H.lr_curve(kind, lr, total, total, **kw)
    if kind == "OneCycleLR":
        got, mom = got
        np.testing.assert_allclose(mom, g["%s_%d_beta1" % (tag, total)], rtol=1e-12, atol=0)
    np.testing.assert_allclose(got, g["%s_%d_lr" % (tag, total)], rtol=1e-9, atol=0)


@pytest.mark.parametrize("total", [200, 333])
@pytest.mark.parametrize("tag", sorted(CASES))
def test_step_schedule_matches_reference(tag, total):
    """The product's scheduler (ao_amd/ptv2/schedule.py; host logic, no GPU) driving a torch optimizer."""
    import torch

    from ao_amd.ptv2.schedule import build_scheduler

    g = load("host_schedules.npz")
    kind, lr, kw = CASES[tag]
    w = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.AdamW([w], lr=lr)
    sched = build_scheduler(dict(type=kind, **kw), opt, total)
    lrs, b1 = [], []
    for s in range(total):
        lrs.append(opt.param_groups[0]["lr"])
        b1.append(opt.param_groups[0]["betas"][0])
        if s == total // 2:  # resume from a checkpoint mid-run
            state = sched.state_dict()
            sched = build_scheduler(dict(type=kind, **kw), opt, total)
            sched.load_state_dict(state)
        sched.step()
    np.testing.assert_allclose(lrs, g["%s_%d_lr" % (tag, total)], rtol=1e-9, atol=0)
    np.testing.assert_allclose(b1, g["%s_%d_beta1" % (tag, total)], rtol=1e-12, atol=0)


def test_build_optimizer_param_groups():
    """optimizer.py:23-46: keyword groups."""
    import torch

    from ao_amd.ptv2.schedule import build_optimizer

    net = torch.nn.ModuleDict(dict(block=torch.nn.Linear(4, 4), head=torch.nn.Linear(4, 2)))
    opt = build_optimizer(dict(type="AdamW", lr=0.01, weight_decay=0.05), net, [dict(keyword="block", lr=0.001)])
    assert isinstance(opt, torch.optim.AdamW) and len(opt.param_groups) == 2
    assert opt.param_groups[0]["lr"] == 0.01 and len(opt.param_groups[0]["params"]) == 2
    assert opt.param_groups[1]["lr"] == 0.001 and len(opt.param_groups[1]["params"]) == 2
    sgd = build_optimizer(dict(type="SGD", lr=0.1, momentum=0.9), net)
    assert isinstance(sgd, torch.optim.SGD)
