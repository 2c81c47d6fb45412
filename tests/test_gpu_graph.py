"""GPU: the model launchers issue their kernel sequence as one hipGraph launch per direction (ao_amd/csrc/graph.hip).

A graph launch must enqueue exactly what eager issue (hipLaunchKernelGGL per kernel, AO_AMD_GRAPH=0) enqueues: the same
kernels with the same arguments in the same order -- so logits, every parameter gradient and the BatchNorm running
statistics are compared BIT for bit, for a sequence of scenes of different sizes (the executable graph of the previous
call is updated in place with the next call's arguments and grids), on the null stream and on a side stream, and with the
sampled kernel timer's time-stamp brackets inside the graph."""
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


def _run(model, scenes):
    out = []
    for data in scenes:
        logits = model(data)
        grads = torch.autograd.grad(F.cross_entropy(logits, data["segment"], ignore_index=-1), list(model.parameters()))
        out.append((logits.detach().clone(), [g.clone() for g in grads]))
    torch.cuda.synchronize()
    return out, {k: v.clone() for k, v in model.state_dict().items()}


@pytest.fixture
def graph_mode(monkeypatch):
    from ao_amd import _lib

    # (the counts below expect every forward prefix as a graph; by default it is issued eagerly when the stream has run dry)
    monkeypatch.setenv("AO_AMD_GRAPH_IDLE_EAGER", "0")

    L = _lib.lib()
    prev = L.ptv2_graph_mode(-1)
    yield L
    L.ptv2_graph_mode(prev)
    L.ptv2_graph_reset()


@pytest.mark.parametrize("pipeline", ["1", "0"])
@pytest.mark.parametrize("tag", ["s3dis", "scannet"])
def test_graph_issue_is_bit_identical_to_eager_over_changing_scene_sizes(graph_mode, monkeypatch, tag, pipeline):
    """pipeline 1 (default): the forward is two graphs -- the level-0 prefix, and the rest behind the geometry's read-backs
    (ao_amd/ptv2/native_model.py) -- plus the backward's; 0: one graph per direction."""
    from ao_amd import _lib

    monkeypatch.setenv("AO_AMD_PIPELINE", pipeline)
    per_call = 3 if pipeline == "1" else 2

    cfg = dict(M.S3DIS_CFG if tag == "s3dis" else M.SCANNET_CFG, drop_path_rate=0.0)
    # six batches, every one with other level sizes (the sizes of the coarse levels are data dependent)
    plan = (([1], 6000), ([2], 9000), ([3, 4], 5000), ([5], 6000), ([6], 14000), ([7], 6100))
    if tag == "scannet":  # (its fourth pooling leaves a handful of points per cloud: two clouds per batch keep every level >= 2 rows)
        plan = tuple((s + [s[0] + 20], n) for s, n in plan)
    scenes = [_data(s, n, cfg) for s, n in plan]
    graph_mode.ptv2_graph_mode(0)
    want, want_state = _run(_model(cfg, seed=3), scenes)
    graph_mode.ptv2_graph_mode(1)
    _lib.graph_stats(reset=True)
    got, got_state = _run(_model(cfg, seed=3), scenes)
    st = _lib.graph_stats()
    assert st["scopes"] == per_call * len(scenes) and st["declined"] == 0, st
    # a ring of three executable graphs per direction: from the fourth call on an existing graph is reused -- updated in
    # place when the node count is unchanged, rebuilt otherwise; both must have happened without an error
    assert st["updated"] + st["instantiated"] == st["scopes"], st
    for (lw, gw), (lg, gg) in zip(want, got):
        assert torch.equal(lw, lg)
        for a, b in zip(gw, gg):
            assert torch.equal(a, b)
    for k in want_state:
        assert torch.equal(want_state[k], got_state[k]), k


def test_graph_update_in_place_for_a_repeated_shape_and_on_a_side_stream(graph_mode):
    from ao_amd import _lib

    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    scenes = [_data([11], 7000, cfg)] * 7
    graph_mode.ptv2_graph_mode(0)
    want, _ = _run(_model(cfg, seed=8), scenes)
    graph_mode.ptv2_graph_mode(1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    _lib.graph_stats(reset=True)
    with torch.cuda.stream(side):
        got, _ = _run(_model(cfg, seed=8), scenes)
    st = _lib.graph_stats()
    # an executable graph is created only when every existing one of the slot is still in flight: at least one per direction,
    # at most the ring; every other call updates one in place (same shape, same node count)
    # (three graphs per call: forward prefix, forward rest, backward)
    assert st["scopes"] == 21 and st["declined"] == 0 and 3 <= st["instantiated"] <= 21 and st["updated"] == 21 - st["instantiated"], st
    # (running statistics differ between the 7 identical calls, the logits therefore too: compared call by call)
    for (lw, gw), (lg, gg) in zip(want, got):
        assert torch.equal(lw, lg)
        for a, b in zip(gw, gg):
            assert torch.equal(a, b)


def test_sampled_kernel_timer_brackets_inside_the_graph(graph_mode):
    """bench.py's roofline leg: with one kernel selected the launchers keep issuing graphs and bracket every stride-th
    launch of that kernel with device time stamps; the all-kernel survey (HIP events) makes the scope decline."""
    from ao_amd import _lib

    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    scenes = [_data([21], 8000, cfg)] * 8
    graph_mode.ptv2_graph_mode(1)
    model = _model(cfg, seed=2)
    _lib.kernel_timer(True)
    _lib.graph_stats(reset=True)
    _run(model, scenes[:1])
    assert _lib.graph_stats()["declined"] == 3  # HIP-event survey: eager (forward prefix, forward rest, backward)
    _lib.kernel_timer(False)
    survey = _lib.kernel_timer_read()
    name = "bn_bwd_finapply_kernel [family]"  # (an id shared by several instantiations; tens of launches per step at this size; the weight gradients are two batched launches)
    assert name in survey and survey[name]["launches"] > 20
    per_step = survey[name]["launches"]
    _lib.kernel_timer(True, only=name, stride=5)
    _lib.graph_stats(reset=True)
    want, _ = _run(model, scenes[1:])
    st = _lib.graph_stats()
    _lib.kernel_timer(False)
    assert st["declined"] == 0 and st["scopes"] == 21, st
    assert st["updated"] >= 9, st  # the brackets of a ring entry keep their positions: graphs are updated, not rebuilt
    rec = _lib.kernel_timer_read()[name]
    assert rec["launches"] >= 7 * (per_step // 5)
    # stamps and events time the same kernel: within a factor (the survey's mean is over all launches, the sample's over a few)
    assert 0.4 < rec["avg_us"] / survey[name]["avg_us"] < 2.5, (rec, survey[name])
    empty = _lib.lib().ptv2_profile_empty_stamp_us(torch.cuda.current_stream().cuda_stream, 50)
    assert 0.0 <= empty < 20.0
    # and the timed launches computed the same thing
    graph_mode.ptv2_graph_mode(0)
    model2 = _model(cfg, seed=2)
    ref, _ = _run(model2, scenes)
    for (lw, gw), (lg, gg) in zip(ref[1:], want):
        assert torch.equal(lw, lg)
        for a, b in zip(gw, gg):
            assert torch.equal(a, b)
