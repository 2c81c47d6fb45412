"""The default forward: geometry built inside PointTransformerV2.forward, pipelined with the network's level-0 prefix
(ao_amd/ptv2/native_model.py::_NativeModel.forward, ao_amd/csrc/model.hip: ptv2_model_forward_prefix / _rest) -- no prefetcher
thread, no `geometry=` batch key: the call the reference trainer makes (pointcept/engines/train_sam_pp2s.py:181,
point_transformer_v2m2_base.py:556-576).  It must enqueue exactly what the one-call forward over a prebuilt geometry does."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ptv2_ref as M

pytestmark = pytest.mark.gpu


def _data(seeds, points, cfg):
    from ao_amd import synth

    b = synth.scene_batch(seeds, point_max=points, in_channels=cfg["in_channels"], num_classes=cfg["num_classes"])
    return {k: torch.from_numpy(v).cuda() for k, v in b.items()}


def _model(cfg, seed):
    import ao_amd.ptv2 as ptv2

    m = ptv2.PointTransformerV2(**cfg).cuda()
    m.load_state_dict(M.init_state(cfg, seed=seed), strict=True)
    return m.train()


def _run(model, data, geometry=None):
    logits = model(data) if geometry is None else model(data, geometry=geometry)
    loss = F.cross_entropy(logits, data["segment"], ignore_index=-1)
    grads = torch.autograd.grad(loss, list(model.parameters()))
    return logits.detach(), grads, {k: v.clone() for k, v in model.state_dict().items()}


@pytest.mark.parametrize("tag,seeds,points", [("s3dis", [1, 2], 6000), ("scannet", [3], 5000), ("s3dis", [4], 40000),
                                              ("scannet", [5, 6], 30000)])
def test_pipelined_forward_equals_the_prebuilt_geometry_forward(monkeypatch, tag, seeds, points):
    from ao_amd.ptv2 import native_model

    cfg = dict(M.S3DIS_CFG if tag == "s3dis" else M.SCANNET_CFG, drop_path_rate=0.0)
    data = _data(seeds, points, cfg)
    res = {}
    for mode in ("pipelined", "prebuilt"):
        model = _model(cfg, seed=23)
        assert native_model.supported(model, data["feat"]) and native_model.pipelined_ok(model)
        geo = model.geometry(data["coord"], data["offset"].int()) if mode == "prebuilt" else None
        for _ in range(2):  # twice: the second call updates the executable graphs of the first
            res[mode] = _run(model, data, geo)
    (lp, gp, sp), (lo, go, so) = res["pipelined"], res["prebuilt"]
    assert torch.equal(lp, lo)
    for (name, _), a, b in zip(model.named_parameters(), gp, go):
        assert torch.equal(a, b), name
    for k in sp:
        assert torch.equal(sp[k], so[k]), k


def test_pipelined_forward_on_changing_batches_and_a_side_stream():
    """Five batches of different sizes back to back (the graphs of prefix and rest are updated, not rebuilt, when the node
    count stays), then the same on a non-default stream: logits equal those of the prebuilt-geometry forward every time."""
    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    a, b = _model(cfg, seed=5), _model(cfg, seed=5)
    for stream in (None, torch.cuda.Stream()):
        for i, points in enumerate((5000, 9000, 7000, 9000, 12000)):
            data = _data([10 + i], points, cfg)
            if stream is not None:
                stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
                la = a(data)
                lb = b(data, geometry=b.geometry(data["coord"], data["offset"].int()))
                F.cross_entropy(la, data["segment"], ignore_index=-1).backward()
                F.cross_entropy(lb, data["segment"], ignore_index=-1).backward()
            torch.cuda.synchronize()
            assert torch.equal(la, lb), (stream is not None, points)
            for (n, p), q in zip(a.named_parameters(), b.parameters()):
                assert torch.equal(p.grad, q.grad), n
            a.zero_grad(set_to_none=True)
            b.zero_grad(set_to_none=True)


def test_pipelined_forward_with_drop_path_is_deterministic_under_a_seed():
    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.3)
    data = _data([8], 8000, cfg)
    outs = []
    for _ in range(2):
        model = _model(cfg, seed=9)
        torch.manual_seed(1234)
        outs.append(_run(model, data))
    assert torch.equal(outs[0][0], outs[1][0])
    assert all(torch.equal(x, y) for x, y in zip(outs[0][1], outs[1][1]))
    assert bool(torch.isfinite(outs[0][0]).all())


def test_eval_mode_and_checkpointing_take_the_one_call_forward(monkeypatch):
    from ao_amd.ptv2 import native_model

    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    data = _data([2], 5000, cfg)
    model = _model(cfg, seed=3)
    ref = model(data).detach()
    model.eval()
    assert not native_model.pipelined_ok(model)
    with torch.no_grad():
        ev = model(data)
    assert ev.shape == ref.shape and bool(torch.isfinite(ev).all())
    monkeypatch.setenv("AO_AMD_PIPELINE", "0")
    model.train()
    assert not native_model.pipelined_ok(model)


@pytest.mark.parametrize("tag,seeds,points", [("s3dis", [1, 2, 3], 9000), ("scannet", [4, 5], 20000), ("s3dis", [6], 120000)])
def test_native_scene_geometry_equals_the_python_sequence(monkeypatch, tag, seeds, points):
    """ptv2_scene_geometry_hip_launcher (csrc/scene.hip: poolings, deeper levels' tables, interpolation tables, inverse tables
    in one native call, carved from one arena) against the same launchers called one by one from python: every table bit for bit."""
    from ao_amd.ptv2 import gva

    cfg = dict(M.S3DIS_CFG if tag == "s3dis" else M.SCANNET_CFG, drop_path_rate=0.0)
    data = _data(seeds, points, cfg)
    model = _model(cfg, seed=1)
    geos = {}
    for mode in ("python", "native"):
        monkeypatch.setenv("AO_AMD_GEOMETRY", mode)
        geos[mode] = model.geometry(data["coord"], data["offset"].int())
    torch.cuda.synchronize()
    a, b = geos["python"], geos["native"]
    assert hasattr(b, "arena") and not hasattr(a, "arena")
    assert len(a.levels) == len(b.levels)
    for i, (la, lb) in enumerate(zip(a.levels, b.levels)):
        for name in ("coord", "offset", "cluster", "order32", "idx_ptr32", "up_idx", "up_weight"):
            x, y = getattr(la, name), getattr(lb, name)
            assert (x is None) == (y is None), (i, name)
            if x is not None:
                assert x.shape == y.shape and torch.equal(x, y), (i, name)
        assert sorted(la.knn) == sorted(lb.knn)
        for k in la.knn:
            assert torch.equal(la.knn[k], lb.knn[k]), (i, k)
            for u, v in zip(gva.inverse_table(la.knn[k]), gva.inverse_table(lb.knn[k])):
                assert torch.equal(u, v), (i, k, "inverse")
            for u, v in zip(gva._pos_moments(gva._HipImpl, la.coord, la.knn[k]), gva._pos_moments(gva._HipImpl, lb.coord, lb.knn[k])):
                assert torch.equal(u, v), (i, k, "moments")
        if la.up_idx is not None:
            ia, ib = gva.inverse_table(la.up_idx), gva.inverse_table(lb.up_idx)
            m = b.levels[i + 1].coord.shape[0]
            assert torch.equal(ia[0][: m + 1], ib[0][: m + 1]) and torch.equal(ia[1], ib[1]), (i, "up inverse")


def test_prefix_issued_eagerly_behind_an_idle_stream_gives_the_same_bits(monkeypatch):
    """A loop that synchronises every step (pointcept's InformationWriter reads the loss back each iteration) finds the stream dry
    at every forward: the prefix is then issued eagerly instead of as a graph (ao_amd/csrc/graph.hip).  Same kernels either way."""
    from ao_amd import _lib

    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    data = _data([12], 9000, cfg)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("AO_AMD_GRAPH_IDLE_EAGER", flag)
        model = _model(cfg, seed=4)
        _lib.graph_stats(reset=True)
        outs = []
        for _ in range(3):
            outs.append(_run(model, data))
            torch.cuda.synchronize()
        res[flag] = (outs, _lib.graph_stats()["scopes"])
    (a, scopes_eager), (b, scopes_graph) = res["1"], res["0"]
    assert scopes_graph == 9 and scopes_eager in (6, 7)  # (the very first call has no earlier graph to look at)
    for (la, ga, sa), (lb, gb, sb) in zip(a, b):
        assert torch.equal(la, lb)
        assert all(torch.equal(x, y) for x, y in zip(ga, gb))


def test_pipelined_forward_takes_a_coord_view_and_an_int64_offset():
    """`coord` handed over as a non-contiguous view (feat[:, :3], how a loader that keeps one (N, 6) tensor would
    pass it) and an int64 `offset` become copy kernels on the caller's stream; the side stream's first pooling has to wait for
    them (the input event is recorded behind the materialisation), so the result equals the contiguous / int32 call's bits."""
    from ao_amd.ptv2 import native_model

    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    data = _data([7, 8], 20000, cfg)
    wide = torch.cat([data["coord"], torch.randn_like(data["coord"])], dim=1)  # (N, 6): coord = a strided view of it
    view = dict(data, coord=wide[:, :3], offset=data["offset"].long())
    assert not view["coord"].is_contiguous() and view["offset"].dtype == torch.int64
    res = {}
    for tag, d in (("plain", data), ("view", view)):
        model = _model(cfg, seed=29)
        assert native_model.supported(model, d["feat"]) and native_model.pipelined_ok(model)
        for _ in range(3):
            res[tag] = _run(model, d)
    assert torch.equal(res["plain"][0], res["view"][0])
    for a, b in zip(res["plain"][1], res["view"][1]):
        assert torch.equal(a, b)
