"""GPU parity: every pointops entry point of libptv2_hip.so (through the python API, i.e. through the
C ABI) against the CPU oracle on identical inputs.  Index outputs must be bit-exact; fp32 feature
outputs within 1e-4 (north_star), gradients of scatter-adds within fp32 reassociation error."""
import numpy as np
import pytest
import torch

from oracle import pointops_ref as P
from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from ao_amd import pointops

    return pointops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cpu(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def check_knn(hp, k, xyz, off, new_xyz=None, new_off=None, pad_with_start=False):
    xd, od = dev(xyz), dev(off)
    if new_xyz is None:
        idx, d2 = hp.knn_query_dist2(k, xd, od, pad_with_start=pad_with_start)
        ridx, rd2 = P.knn_query_raw(k, cpu(xyz), cpu(off), pad_with_start=pad_with_start)
    else:
        idx, d2 = hp.knn_query_dist2(k, xd, od, dev(new_xyz), dev(new_off), pad_with_start=pad_with_start)
        ridx, rd2 = P.knn_query_raw(k, cpu(xyz), cpu(off), cpu(new_xyz), cpu(new_off), pad_with_start=pad_with_start)
    torch.cuda.synchronize()
    bad = (idx.cpu() != ridx).any(1).nonzero().flatten()
    assert bad.numel() == 0, "k=%d: %d/%d rows differ, first %s: hip %s ref %s" % (
        k, bad.numel(), ridx.shape[0], bad[:3].tolist(), idx.cpu()[bad[:1]].tolist(), ridx[bad[:1]].tolist())
    assert torch.equal(d2.cpu(), rd2), "squared distances differ bitwise"


def test_knn_golden_vectors(hp, golden):
    g = golden("knn_random.npz")
    for k in (1, 3, 8, 16):
        idx, dist = hp.knn_query(k, dev(g["xyz"]), dev(g["offset"]))
        assert np.array_equal(idx.cpu().numpy(), g["idx_k%d" % k])
        np.testing.assert_allclose(dist.cpu().numpy(), g["dist_k%d" % k], rtol=2e-7, atol=0)
    idx, dist = hp.knn_query(3, dev(g["xyz"]), dev(g["offset"]), dev(g["new_xyz"]), dev(g["new_offset"]))
    assert np.array_equal(idx.cpu().numpy(), g["cross_idx_k3"])
    g = golden("knn_lattice.npz")
    for k in (4, 16):
        idx, _ = hp.knn_query(k, dev(g["xyz"]), dev(g["offset"]))
        assert np.array_equal(idx.cpu().numpy(), g["idx_k%d" % k])
    g = golden("knn_short.npz")
    idx, dist = hp.knn_query(16, dev(g["xyz"]), dev(g["offset"]))
    assert np.array_equal(idx.cpu().numpy(), g["idx_k16"])
    np.testing.assert_allclose(dist.cpu().numpy(), g["dist_k16"], rtol=2e-7)


@pytest.mark.parametrize("k", [1, 2, 3, 5, 8, 12, 16, 20, 32, 40, 128])
def test_knn_random_clouds_all_k(hp, k):
    xyz = synth.random_cloud(3000, seed=k)
    off = np.array([700, 701, 1900, 3000], np.int32)  # includes a 1-point cloud
    check_knn(hp, k, xyz, off)
    q = synth.random_cloud(1111, seed=100 + k, scale=2.5)  # some queries outside the bbox
    qoff = np.array([300, 300, 800, 1111], np.int32)  # includes an empty query segment
    check_knn(hp, k, xyz, off, q, qoff)


def test_knn_ties_duplicates_degenerate(hp):
    lat = synth.lattice_cloud(9, 7, 5, 0.125)
    check_knn(hp, 16, lat, np.array([100, lat.shape[0]], np.int32))
    check_knn(hp, 3, lat, np.array([lat.shape[0]], np.int32), synth.lattice_cloud(4, 4, 4, 0.25) + 0.0625,
              np.array([64], np.int32))
    dup = np.repeat(synth.random_cloud(300, seed=9), 3, axis=0)  # every point three times
    check_knn(hp, 8, dup, np.array([dup.shape[0]], np.int32))
    plane = synth.random_cloud(2000, seed=10)
    plane[:, 2] = 0.5  # planar cloud: degenerate bbox axis
    check_knn(hp, 16, plane, np.array([2000], np.int32))
    line = np.zeros((500, 3), np.float32)
    line[:, 0] = np.linspace(0, 1, 500, dtype=np.float32)
    check_knn(hp, 4, line, np.array([500], np.int32))
    same = np.ones((50, 3), np.float32)  # zero-extent cloud
    check_knn(hp, 5, same, np.array([50], np.int32))
    short = synth.random_cloud(40, seed=3)
    check_knn(hp, 16, short, np.array([5, 12, 40], np.int32))
    check_knn(hp, 16, short, np.array([5, 12, 40], np.int32), pad_with_start=True)


def test_knn_room_scene_full_size(hp):
    """S3DIS-shaped cloud at BASELINE size: bit-exact vs. the multi-threaded oracle on a query subset,
    plus size-independent properties on the whole output."""
    b = synth.scene_batch([0], point_max=80000)
    xyz, off = b["coord"], b["offset"]
    n = xyz.shape[0]
    idx, d2 = hp.knn_query_dist2(16, dev(xyz), dev(off))
    idx, d2 = idx.cpu(), d2.cpu()
    assert (idx[:, 0] == torch.arange(n)).all() and (d2[:, 0] == 0).all()
    assert (d2[:, 1:] >= d2[:, :-1]).all() and (idx >= 0).all() and (idx < n).all()
    sub = np.sort(np.random.default_rng(0).choice(n, 4000, replace=False))
    ridx, rd2 = P.knn_query_raw(16, cpu(xyz), cpu(off), cpu(xyz[sub]), torch.tensor([len(sub)], dtype=torch.int32), mt=True)
    assert torch.equal(idx[sub], ridx) and torch.equal(d2[sub], rd2)
    # cross query (interpolation shape): k=3 fine -> coarse
    coarse = xyz[::6].copy()
    cidx, cd2 = hp.knn_query_dist2(3, dev(coarse), dev(np.array([coarse.shape[0]], np.int32)), dev(xyz), dev(off))
    ridx, rd2 = P.knn_query_raw(3, cpu(coarse), torch.tensor([coarse.shape[0]], dtype=torch.int32), cpu(xyz[sub]),
                                torch.tensor([len(sub)], dtype=torch.int32), mt=True)
    assert torch.equal(cidx.cpu()[sub], ridx) and torch.equal(cd2.cpu()[sub], rd2)


def test_knn_queries_sharing_one_cell_grid(hp):
    """KnnGrid: the interpolation table (k = 3, queries from a finer cloud), two self tables and a tie-heavy lattice over ONE
    grid build give the bits of independent calls; a fifth query, other source points or a larger query set rebuild it."""
    from ao_amd.pointops.query import KnnGrid

    lat = synth.lattice_cloud(12, 10, 8, 0.125)  # ties everywhere: every query of the shared grid fills its own re-run list
    src = np.concatenate([synth.random_cloud(6000, seed=21), lat]).astype(np.float32)
    off = np.array([2500, 6000, src.shape[0]], np.int32)
    fine = np.concatenate([synth.random_cloud(15000, seed=22, scale=1.1), lat + 0.0625]).astype(np.float32)
    foff = np.array([7000, 15000, fine.shape[0]], np.int32)
    x, o, q, qo = dev(src), dev(off), dev(fine), dev(foff)
    grid = KnnGrid()
    calls = [(3, q, qo), (16, None, None), (8, None, None), (1, q, qo), (16, None, None)]
    modes = []
    for k, nq, nqo in calls:
        got = hp.knn_query_dist2(k, x, o, nq, nqo, grid=grid)
        modes.append(grid.used)
        ref = hp.knn_query_dist2(k, x, o, nq, nqo)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), k
    assert modes == [1, 2, 3, 4, 1]  # the fifth query built the grid again
    ridx, rd2 = P.knn_query_raw(16, cpu(src), cpu(off), cpu(src), cpu(off), mt=True)
    got = hp.knn_query_dist2(16, x, o, grid=grid)
    assert torch.equal(got[0].cpu(), ridx) and torch.equal(got[1].cpu(), rd2)
    other = dev(src[::-1].copy())  # other points (another address): not the grid in the workspace
    got = hp.knn_query_dist2(8, other, o, grid=grid)
    assert grid.used == 1
    ref = hp.knn_query_dist2(8, other, o)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    big = dev(np.concatenate([fine, fine + 0.01, fine - 0.01]).astype(np.float32))  # more queries than the workspace was sized for
    bo = dev(np.array([3 * 7000, 3 * 15000, 3 * fine.shape[0]], np.int32))
    got = hp.knn_query_dist2(3, other, o, big, bo, grid=grid)
    ref = hp.knn_query_dist2(3, other, o, big, bo)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


def test_knn_multi_scene_batch(hp):
    b = synth.scene_batch([1, 2, 3], point_max=20000)
    check_knn(hp, 16, b["coord"], b["offset"])


def test_fps(hp, golden):
    g = golden("fps.npz")
    idx = hp.farthest_point_sampling(dev(g["xyz"]), dev(g["offset"]), dev(g["new_offset"]))
    assert np.array_equal(idx.cpu().numpy(), g["idx"])
    idx = hp.farthest_point_sampling(dev(g["dup_xyz"]), dev(g["dup_offset"]), dev(g["dup_new_offset"]))
    assert np.array_equal(idx.cpu().numpy(), g["dup_idx"])
    # > 12 K points per cloud exercises the streamed tail; lattice part exercises ties at B = 1024
    xyz = np.concatenate([synth.random_cloud(20000, seed=5), synth.lattice_cloud(20, 20, 10, 0.1)])
    off = np.array([20000, 24000], np.int32)
    noff = np.array([600, 900], np.int32)
    idx = hp.farthest_point_sampling(dev(xyz), dev(off), dev(noff))
    ref = P.farthest_point_sampling(cpu(xyz), cpu(off), cpu(noff))
    assert torch.equal(idx.cpu(), ref)
    small = synth.random_cloud(37, seed=6)  # B = 32 < 1024, 2 rounds of virtual threads
    idx = hp.farthest_point_sampling(dev(small), dev(np.array([37], np.int32)), dev(np.array([37], np.int32)))
    ref = P.farthest_point_sampling(cpu(small), torch.tensor([37], dtype=torch.int32), torch.tensor([37], dtype=torch.int32))
    assert torch.equal(idx.cpu(), ref) and sorted(idx.cpu().tolist()) == list(range(37))


def test_grouping(hp, golden):
    g = golden("grouping.npz")
    feat = dev(g["feat"]).requires_grad_(True)
    o = hp.grouping(dev(g["idx_m1"]), feat, dev(g["xyz"]), with_xyz=True)
    np.testing.assert_allclose(o.detach().cpu().numpy(), g["out_xyz"], rtol=0, atol=1e-6)
    (gf,) = torch.autograd.grad(o, feat, dev(g["grad_out"]))
    np.testing.assert_allclose(gf.cpu().numpy(), g["grad_feat"], rtol=1e-5, atol=1e-5)
    o2 = hp.grouping2(feat, dev(g["idx"]))
    assert np.array_equal(o2.detach().cpu().numpy(), g["out2"])
    (gf2,) = torch.autograd.grad(o2, feat, dev(g["grad_out2"]))
    np.testing.assert_allclose(gf2.cpu().numpy(), g["grad_feat2"], rtol=1e-5, atol=1e-5)
    for c in (1, 3, 6, 48, 50):  # vector and scalar paths
        inp = torch.randn(500, c, device="cuda")
        idx = torch.randint(-1, 500, (333, 7), device="cuda", dtype=torch.int32)
        out = hp.grouping2(inp, idx)
        ref = torch.cat([inp, inp.new_zeros(1, c)])[idx.long()]
        assert torch.equal(out, ref)


def test_interpolation(hp, golden):
    g = golden("interpolation.npz")
    feat = dev(g["feat"]).requires_grad_(True)
    args = (dev(g["xyz"]), dev(g["new_xyz"]), feat, dev(g["offset"]), dev(g["new_offset"]))
    for fn in (hp.interpolation, hp.interpolation2):
        o = fn(*args)
        np.testing.assert_allclose(o.detach().cpu().numpy(), g["out"], rtol=1e-5, atol=1e-5)
        (gf,) = torch.autograd.grad(o, feat, dev(g["grad_out"]))
        np.testing.assert_allclose(gf.cpu().numpy(), g["grad_feat"], rtol=1e-4, atol=1e-4)
    # coarse segment shorter than k: the reference wraps index -1 to the last row (interpolation.py:21)
    cx, co = synth.random_cloud(12, seed=1), np.array([2, 12], np.int32)
    fx, fo = synth.random_cloud(30, seed=2), np.array([10, 30], np.int32)
    cf = torch.randn(12, 5)
    ref = P.interpolation(cpu(cx), cpu(fx), cf, cpu(co), cpu(fo))
    out = hp.interpolation(dev(cx), dev(fx), cf.cuda(), dev(co), dev(fo))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def test_subtraction_aggregation_attention(hp, golden):
    g = golden("subtraction.npz")
    a, b = dev(g["in1"]).requires_grad_(True), dev(g["in2"]).requires_grad_(True)
    out = hp.subtraction(a, b, dev(g["idx"]))
    assert np.array_equal(out.detach().cpu().numpy(), g["out"])
    ga, gb = torch.autograd.grad(out, (a, b), dev(g["grad_out"]))
    np.testing.assert_allclose(ga.cpu().numpy(), g["grad_in1"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gb.cpu().numpy(), g["grad_in2"], rtol=1e-5, atol=1e-5)

    g = golden("aggregation.npz")
    inp, pos, w = (dev(g[k]).requires_grad_(True) for k in ("input", "position", "weight"))
    out = hp.aggregation(inp, pos, w, dev(g["idx"]))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=1e-5, atol=1e-5)
    gi, gp, gw = torch.autograd.grad(out, (inp, pos, w), dev(g["grad_out"]))
    np.testing.assert_allclose(gi.cpu().numpy(), g["grad_input"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gp.cpu().numpy(), g["grad_position"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(gw.cpu().numpy(), g["grad_weight"], rtol=1e-4, atol=1e-4)

    g = golden("attention.npz")
    q, k, v = (dev(g[n]).requires_grad_(True) for n in ("query", "key", "value"))
    tgt, ref = dev(g["index_target"]), dev(g["index_refer"])
    rel = hp.attention_relation_step(q, k, dev(g["weight"]), tgt, ref)
    np.testing.assert_allclose(rel.detach().cpu().numpy(), g["relation"], rtol=1e-5, atol=1e-5)
    gq, gk = torch.autograd.grad(rel, (q, k), dev(g["grad_relation"]))
    np.testing.assert_allclose(gq.cpu().numpy(), g["grad_query"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gk.cpu().numpy(), g["grad_key"], rtol=1e-4, atol=1e-4)
    aw = dev(g["attn"]).requires_grad_(True)
    fu = hp.attention_fusion_step(aw, v, tgt, ref)
    np.testing.assert_allclose(fu.detach().cpu().numpy(), g["fused"], rtol=1e-5, atol=1e-5)
    gaw, gv = torch.autograd.grad(fu, (aw, v), dev(g["grad_fused"]))
    np.testing.assert_allclose(gaw.cpu().numpy(), g["grad_attn"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gv.cpu().numpy(), g["grad_value"], rtol=1e-4, atol=1e-4)


def test_pointops2_spellings(hp):
    from ao_amd.pointops2 import pointops as p2

    xyz = synth.random_cloud(40, seed=3)
    off = np.array([5, 12, 40], np.int32)
    idx, dist = p2.knnquery(16, dev(xyz), None, dev(off), None)
    ridx, rd2 = P.knn_query_raw(16, cpu(xyz), cpu(off), pad_with_start=True)
    assert torch.equal(idx.cpu(), ridx)
    fi = p2.furthestsampling(dev(xyz), dev(off), dev(np.array([2, 5, 15], np.int32)))
    assert torch.equal(fi.cpu(), P.farthest_point_sampling(cpu(xyz), cpu(off), torch.tensor([2, 5, 15], dtype=torch.int32)))
    feat = torch.randn(40, 8, device="cuda")
    out = p2.queryandgroup(4, dev(xyz), dev(xyz), feat, None, dev(off), dev(off))
    assert out.shape == (40, 4, 11)


def test_dropin_import_path():
    import os
    import sys

    from tests.conftest import ROOT

    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    import pointops

    xyz = dev(synth.random_cloud(100, seed=0))
    idx, dist = pointops.knn_query(4, xyz, dev(np.array([100], np.int32)))
    assert idx.dtype == torch.int32 and idx.shape == (100, 4) and dist.dtype == torch.float32


def test_fps_cooperative_multi_workgroup(hp):
    """Clouds above 8 K points take the multi-workgroup path (granule hand-off per sample); bit-exact vs oracle,
    including exact ties (lattice) and clouds of very different sizes in one call."""
    xyz = np.concatenate([synth.random_cloud(30000, seed=11), synth.lattice_cloud(24, 24, 16, 0.05),
                          synth.random_cloud(700, seed=12)])
    off = np.array([30000, 30000 + 9216, 30000 + 9216 + 700], np.int32)
    noff = np.array([400, 700, 760], np.int32)
    idx = hp.farthest_point_sampling(dev(xyz), dev(off), dev(noff))
    ref = P.farthest_point_sampling(cpu(xyz), cpu(off), cpu(noff))
    assert torch.equal(idx.cpu(), ref)
    # > 2048 samples: the granule's sample tag wraps
    b = synth.scene_batch([2], point_max=20000)
    n = b["coord"].shape[0]
    idx = hp.farthest_point_sampling(dev(b["coord"]), dev(b["offset"]), dev(np.array([2300], np.int32)))
    ref = P.farthest_point_sampling(cpu(b["coord"]), cpu(b["offset"]), torch.tensor([2300], dtype=torch.int32))
    assert torch.equal(idx.cpu(), ref) and len(set(idx.cpu().tolist())) == 2300


@pytest.mark.parametrize("local", ["1", "0"])
def test_fps_exchange_through_one_xcd_and_through_memory(hp, local):
    """The cooperative kernel's two exchanges -- all workgroups of a cloud on one XCD (granules through that XCD's L2) and the
    device-scope one (AO_AMD_FPS_LOCAL=0) -- on 1, 3 and 11 clouds (more clouds than XCDs: two teams per XCD), sizes from one
    workgroup's worth to the bench scene; bit-exact vs the oracle."""
    import subprocess
    import sys

    code = r"""
import numpy as np, torch
from ao_amd import pointops
from oracle import pointops_ref as P
from tests import synth
rng = np.random.default_rng(5)
for sizes, frac in [([120000], 40), ([30000, 9000, 45000], 30), ([9000 + 1500 * i for i in range(11)], 25)]:
    xyz = np.concatenate([synth.random_cloud(n, seed=100 + i) for i, n in enumerate(sizes)]).astype(np.float32)
    off = np.cumsum(sizes).astype(np.int32)
    noff = np.cumsum([max(2, n // frac // 10) for n in sizes]).astype(np.int32)
    idx = pointops.farthest_point_sampling(torch.from_numpy(xyz).cuda(), torch.from_numpy(off).cuda(), torch.from_numpy(noff).cuda())
    ref = P.farthest_point_sampling(torch.from_numpy(xyz), torch.from_numpy(off), torch.from_numpy(noff))
    assert torch.equal(idx.cpu(), ref), sizes
print("ok")
"""
    import os

    env = dict(os.environ, AO_AMD_FPS_LOCAL=local)  # (the launcher reads the switch per call; a fresh process keeps it simple)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_empty_inputs(hp):
    """Zero queries / zero rows: every op returns an empty result of the right shape and dtype (the reference launches
    zero-size grids; here the launchers return before launching)."""
    xyz = dev(synth.random_cloud(100, seed=1))
    off = dev(np.array([100], np.int32))
    none = torch.zeros((0, 3), device="cuda")
    noff = dev(np.array([0], np.int32))
    idx, dist = hp.knn_query(4, xyz, off, none, noff)
    assert idx.shape == (0, 4) and idx.dtype == torch.int32 and dist.shape == (0, 4)
    feat = torch.randn(100, 8, device="cuda", requires_grad=True)
    g = hp.grouping(idx, feat, xyz, none, with_xyz=True)
    assert g.shape == (0, 4, 11)
    g.sum().backward()
    assert feat.grad is not None and float(feat.grad.abs().sum()) == 0.0
    out = hp.interpolation(xyz, none, feat.detach(), off, noff)
    assert out.shape == (0, 8)
    sub = hp.subtraction(torch.zeros((0, 8), device="cuda"), torch.zeros((0, 8), device="cuda"), torch.zeros((0, 4), dtype=torch.int32, device="cuda"))
    assert sub.shape == (0, 4, 8)
