"""CPU: the oracle (oracle/) against the golden fixtures and against its own stated properties.

Fixtures come from the reference's python wrappers executed in the build
container (tests/golden/make_golden.py).  Ops whose reference implementation is
pure torch (`grouping`, `interpolation`) are genuinely pinned by them; for ops
that bottom out in a CUDA kernel the fixture is an oracle regression vector.
"""
import numpy as np
import pytest
import torch

from oracle import pointops_ref as P
from tests import synth


def t(a):
    return torch.from_numpy(np.asarray(a))


def brute_knn_sorted(k, xyz, offset, new_xyz, new_offset):
    """Independent statement of 'k smallest by (d2, ascending)' with the pinned d2 expression."""
    xyz, new_xyz = xyz.astype(np.float32), new_xyz.astype(np.float32)
    m = new_xyz.shape[0]
    idx = np.full((m, k), -1, np.int32)
    d2o = np.full((m, k), 1e10, np.float32)
    starts = np.concatenate([[0], offset[:-1]])
    nstarts = np.concatenate([[0], new_offset[:-1]])
    for b in range(len(offset)):
        pts = xyz[starts[b]:offset[b]]
        for q in range(nstarts[b], new_offset[b]):
            dx = (new_xyz[q, 0] - pts[:, 0]).astype(np.float32)
            dy = (new_xyz[q, 1] - pts[:, 1]).astype(np.float32)
            dz = (new_xyz[q, 2] - pts[:, 2]).astype(np.float32)
            # fma(dz,dz,fma(dx,dx,dy*dy)) evaluated exactly in float64 then rounded once per fma
            inner = (dx.astype(np.float64) * dx + (dy * dy).astype(np.float32)).astype(np.float32)
            d2 = (dz.astype(np.float64) * dz + inner).astype(np.float32)
            order = np.argsort(d2, kind="stable")[:k]
            idx[q, : len(order)] = order + starts[b]
            d2o[q, : len(order)] = d2[order]
    return idx, d2o


def test_knn_golden_and_sorted_property(golden):
    g = golden("knn_random.npz")
    xyz, off = t(g["xyz"]), t(g["offset"])
    for k in (1, 3, 8, 16):
        idx, dist = P.knn_query(k, xyz, off)
        assert np.array_equal(idx.numpy(), g["idx_k%d" % k])
        assert np.array_equal(dist.numpy(), g["dist_k%d" % k])
        bi, bd = brute_knn_sorted(k, g["xyz"], g["offset"], g["xyz"], g["offset"])
        assert np.array_equal(idx.numpy(), bi)  # no ties in random data -> plain sorted top-k
        assert np.array_equal(dist.numpy(), torch.sqrt(t(bd)).numpy())
        assert (idx[:, 0].numpy() == np.arange(xyz.shape[0])).all()  # self is its own nearest
    idx, dist = P.knn_query(3, xyz, off, t(g["new_xyz"]), t(g["new_offset"]))
    assert np.array_equal(idx.numpy(), g["cross_idx_k3"])
    assert np.array_equal(dist.numpy(), g["cross_dist_k3"])
    # queries never see the other cloud
    assert (idx[:300] < 900).all() and (idx[300:] >= 900).all()


def test_knn_lattice_ties_and_short_segments(golden):
    g = golden("knn_lattice.npz")
    for k in (4, 16):
        idx, dist = P.knn_query(k, t(g["xyz"]), t(g["offset"]))
        assert np.array_equal(idx.numpy(), g["idx_k%d" % k])
        # distances are still the sorted k smallest even when the index choice is heap-defined
        _, bd = brute_knn_sorted(k, g["xyz"], g["offset"], g["xyz"], g["offset"])
        assert np.array_equal(dist.numpy(), torch.sqrt(t(bd)).numpy())
    g = golden("knn_short.npz")
    idx, dist = P.knn_query(16, t(g["xyz"]), t(g["offset"]))
    assert np.array_equal(idx.numpy(), g["idx_k16"])
    assert (idx[:5, 5:] == -1).all() and (idx[:5, :5] >= 0).all()
    assert (idx[5:12, 7:] == -1).all()
    assert np.allclose(dist[:5, 5:].numpy(), 1e5)  # sqrt(1e10) placeholders (query.py:24)
    idx2, _ = P.knn_query_raw(16, t(g["xyz"]), t(g["offset"]), pad_with_start=True)
    assert (idx2[:5, 5:] == 0).all() and (idx2[5:12, 7:] == 5).all()  # pointops2 variant pads with `start`


def test_knn_tie_rule_is_sufficient():
    """if the k+1 smallest d2 are pairwise distinct the heap result is the sorted top-k,
    whatever ties exist further out.  Integer lattices with jitter on a few points."""
    rng = np.random.default_rng(0)
    checked = 0
    for trial in range(40):
        n = int(rng.integers(20, 120))
        pts = rng.integers(0, 4, (n, 3)).astype(np.float32)
        off = np.array([n], np.int32)
        k = int(rng.integers(1, 9))
        idx, d = P.knn_query_raw(k, t(pts), t(off))
        bi, bd = brute_knn_sorted(k + 1, pts, off, pts, off)
        for q in range(n):
            vals = bd[q][bd[q] < 1e10]
            if len(np.unique(vals)) == len(vals):
                assert np.array_equal(idx[q].numpy(), bi[q, :k]), (trial, q)
                checked += 1
    assert checked > 50


def test_fps_golden_and_tie_rule(golden):
    g = golden("fps.npz")
    idx = P.farthest_point_sampling(t(g["xyz"]), t(g["offset"]), t(g["new_offset"]))
    assert np.array_equal(idx.numpy(), g["idx"])
    assert idx[0] == 0 and idx[375] == 1500  # each cloud starts from its first point
    assert len(set(idx[:375].tolist())) == 375 and (idx[:375] < 1500).all()
    idx = P.farthest_point_sampling(t(g["dup_xyz"]), t(g["dup_offset"]), t(g["dup_new_offset"]))
    assert np.array_equal(idx.numpy(), g["dup_idx"])
    # closed form of the tie rule: argmin (bitreverse((k-start) mod B), k) among maximal tmp
    xyz = g["dup_xyz"]
    n = xyz.shape[0]
    B = 1 << int(np.floor(np.log2(n)))
    bits = int(np.log2(B))
    tmp = np.full(n, 1e10, np.float32)
    old, sel = 0, [0]
    for _ in range(1, 100):
        d = ((xyz - xyz[old]) ** 2).astype(np.float32)
        dd = (d[:, 2].astype(np.float64) + (d[:, 0].astype(np.float64) + d[:, 1]).astype(np.float32)).astype(np.float32)
        tmp = np.minimum(tmp, dd)
        cand = np.nonzero(tmp == tmp.max())[0]
        key = [(int(format(int(c % B), "0%db" % bits)[::-1], 2), int(c)) for c in cand]
        old = min(key)[1]
        sel.append(old)
    assert sel == g["dup_idx"].tolist()


def test_grouping_interpolation_match_reference_python(golden):
    g = golden("grouping.npz")
    feat = t(g["feat"]).requires_grad_(True)
    o = P.grouping(t(g["idx_m1"]), feat, t(g["xyz"]), with_xyz=True)
    assert np.array_equal(o.detach().numpy(), g["out_xyz"])
    (gf,) = torch.autograd.grad(o, feat, t(g["grad_out"]))
    np.testing.assert_allclose(gf.numpy(), g["grad_feat"], rtol=1e-5, atol=1e-5)
    o2 = P.grouping2(feat, t(g["idx"]))
    assert np.array_equal(o2.detach().numpy(), g["out2"])
    (gf2,) = torch.autograd.grad(o2, feat, t(g["grad_out2"]))
    np.testing.assert_allclose(gf2.numpy(), g["grad_feat2"], rtol=1e-5, atol=1e-5)

    g = golden("interpolation.npz")
    feat = t(g["feat"]).requires_grad_(True)
    o = P.interpolation(t(g["xyz"]), t(g["new_xyz"]), feat, t(g["offset"]), t(g["new_offset"]))
    np.testing.assert_allclose(o.detach().numpy(), g["out"], rtol=1e-6, atol=1e-6)
    (gf,) = torch.autograd.grad(o, feat, t(g["grad_out"]))
    np.testing.assert_allclose(gf.numpy(), g["grad_feat"], rtol=1e-5, atol=1e-5)
    o2 = P.interpolation2(t(g["xyz"]), t(g["new_xyz"]), feat, t(g["offset"]), t(g["new_offset"]))
    np.testing.assert_allclose(o2.detach().numpy(), g["out"], rtol=1e-5, atol=1e-5)
    (gf2,) = torch.autograd.grad(o2, feat, t(g["grad_out"]))
    np.testing.assert_allclose(gf2.numpy(), g["grad_feat"], rtol=1e-5, atol=1e-5)


def test_sub_agg_attention_against_torch_definitions(golden):
    g = golden("subtraction.npz")
    idx = t(g["idx"]).long()
    a, b = t(g["in1"]).requires_grad_(True), t(g["in2"]).requires_grad_(True)
    ref = a.unsqueeze(1) - b[idx]
    out = P.subtraction(a, b, t(g["idx"]))
    assert np.array_equal(out.detach().numpy(), g["out"]) and torch.equal(out, ref)
    ga, gb = torch.autograd.grad(ref, (a, b), t(g["grad_out"]))
    np.testing.assert_allclose(g["grad_in1"], ga.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g["grad_in2"], gb.numpy(), rtol=1e-5, atol=1e-5)

    g = golden("aggregation.npz")
    idx = t(g["idx"]).long()
    inp, pos, w = (t(g[k]).requires_grad_(True) for k in ("input", "position", "weight"))
    n, k, c = pos.shape
    ref = ((inp[idx] + pos) * w.repeat(1, 1, c // w.shape[-1])).sum(1)
    out = P.aggregation(inp, pos, w, t(g["idx"]))
    np.testing.assert_allclose(out.detach().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g["out"], ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    gi, gp, gw = torch.autograd.grad(ref, (inp, pos, w), t(g["grad_out"]))
    np.testing.assert_allclose(g["grad_input"], gi.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(g["grad_position"], gp.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g["grad_weight"], gw.numpy(), rtol=1e-4, atol=1e-4)

    g = golden("attention.npz")
    q, kk, v = (t(g[k]).requires_grad_(True) for k in ("query", "key", "value"))
    wv, tg, rf = t(g["weight"]), t(g["index_target"]).long(), t(g["index_refer"]).long()
    ref = (q[tg] * kk[rf] * wv).sum(-1)
    out = P.attention_relation_step(q, kk, wv, t(g["index_target"]), t(g["index_refer"]))
    np.testing.assert_allclose(out.detach().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g["relation"], ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    gq, gk = torch.autograd.grad(ref, (q, kk), t(g["grad_relation"]))
    np.testing.assert_allclose(g["grad_query"], gq.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(g["grad_key"], gk.numpy(), rtol=1e-4, atol=1e-4)
    aw = t(g["attn"]).requires_grad_(True)
    ref = torch.zeros_like(v).index_add(0, tg, aw.unsqueeze(-1) * v[rf])
    np.testing.assert_allclose(g["fused"], ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    gaw, gv = torch.autograd.grad(ref, (aw, v), t(g["grad_fused"]))
    np.testing.assert_allclose(g["grad_attn"], gaw.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(g["grad_value"], gv.numpy(), rtol=1e-4, atol=1e-4)


def test_synth_scene_shapes():
    b = synth.scene_batch([0], point_max=6000)
    assert b["coord"].shape == (6000, 3) and b["feat"].shape == (6000, 6)
    assert b["offset"].tolist() == [6000] and b["segment"].min() == -1 and b["segment"].max() == 12
    # voxel dedupe at 0.04 m: nearest-neighbour spacing is of that order and no duplicates survive
    _, d = P.knn_query(2, t(b["coord"]), t(b["offset"]))
    assert d[:, 1].min() > 0 and 0.02 < float(d[:, 1].median()) < 0.06
