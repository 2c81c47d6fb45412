"""GPU: attention dropout (GroupedVectorAttention.attn_drop, point_transformer_v2m2_base.py:101,122) inside the fused kernels.

The fused path keeps no mask: every softmax weight's factor (0 or 1/(1-p)) is a hash of (element index, per-Block per-step
seed) that the forward softmax kernels and the backward point kernel both evaluate (ao_amd/csrc/gva_common.h:
ptv2_drop_factor).  The tests give the literal op sequence (AO_AMD_GVA=unfused) the SAME factors through
gva.attn_drop_mask, the torch statement of that hash, and compare output, every gradient and the running statistics."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle.ptv2_ref as M

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


ZERO_GRAD = ("linear_q.0.bias", "linear_k.0.bias", "linear_v.bias", "linear_p_bias.0.bias", "linear_p_bias.3.bias",
             "weight_encoding.0.bias", "weight_encoding.3.bias")


def _grad_close(name, a, b):
    """The bounds of tests/test_gpu_block.py: two fp32 evaluations with different summation orders (1e-2: a 1e-6 difference
    in a pre-activation flips a ReLU mask and moves a whole term of a BatchNorm parameter gradient); biases whose effect a
    training-mode BatchNorm / the softmax's shift invariance removes have an exactly zero gradient that both sides compute
    as summation noise.  A wrong mask (another element order, another seed) shows at O(1) in every gradient."""
    floor = 5e-3 if any(name.endswith(z) for z in ZERO_GRAD) else 1e-4
    return rel(a, b) < 1e-2 or float((a - b).abs().max()) < floor


def _neighbours(n, k, seed):
    from ao_amd import pointops

    g = torch.Generator().manual_seed(seed)
    coord = (torch.rand(n, 3, generator=g) * 4).cuda()
    offset = torch.tensor([n // 2, n], dtype=torch.int32).cuda()
    idx, _ = pointops.knn_query(k, coord, offset)
    return coord, idx


def test_mask_statement():
    """gva.attn_drop_mask: values in {0, 1/(1-p)}, keep fraction 1-p, different seeds independent, p=0 keeps everything."""
    from ao_amd.ptv2 import gva

    n, k, g = 5000, 16, 12
    for p in (0.1, 0.3, 0.5):
        m = gva.attn_drop_mask(1234, n, k, g, p, "cuda")
        vals = torch.unique(m).cpu().numpy()
        np.testing.assert_allclose(vals, [0.0, 1.0 / (1.0 - p)], rtol=1e-6)
        keep = float((m > 0).float().mean())
        assert abs(keep - (1 - p)) < 4 * (p * (1 - p) / (n * k * g)) ** 0.5 + 1e-4, (p, keep)
        # per group / per slot too (the hash must not line up with the (slot, group) layout)
        assert float(((m > 0).float().mean((0, 1)) - (1 - p)).abs().max()) < 0.01
        assert float(((m > 0).float().mean((0, 2)) - (1 - p)).abs().max()) < 0.01
        m2 = gva.attn_drop_mask(1235, n, k, g, p, "cuda")
        both = float(((m > 0) & (m2 > 0)).float().mean())
        assert abs(both - (1 - p) ** 2) < 0.005
    assert bool((gva.attn_drop_mask(7, 100, 8, 6, 0.0, "cuda") == 1).all())
    assert bool((gva.attn_drop_mask(7, 100, 8, 6, 1.0, "cuda") == 0).all())


@pytest.mark.parametrize("c,g,k,n", [(48, 6, 16, 3001), (96, 12, 16, 2000), (96, 12, 8, 2000), (192, 24, 16, 1500),
                                     (384, 48, 16, 700), (512, 64, 16, 300), (48, 6, 4, 900)])
@pytest.mark.parametrize("p", [0.2, 0.5])
def test_attention_dropout_fused_equals_masked_literal(monkeypatch, c, g, k, n, p):
    import ao_amd.ptv2 as ptv2
    from ao_amd.ptv2 import gva

    assert gva.dropout_supported(c, g, k)
    torch.manual_seed(c + k)
    attn = ptv2.GroupedVectorAttention(c, g, attn_drop_rate=p).cuda().train()
    coord, idx = _neighbours(n, k, seed=n)
    idx[5, k - 1] = -1  # (an empty slot, as a short segment leaves)
    feat0 = torch.randn(n, c, device="cuda")
    gout = torch.randn(n, c, device="cuda")
    state0 = {kk: v.clone() for kk, v in attn.state_dict().items()}
    res = {}
    for mode in ("fused", "unfused"):
        monkeypatch.setenv("AO_AMD_GVA", mode)
        attn.load_state_dict(state0)
        feat = feat0.clone().requires_grad_(True)
        torch.manual_seed(99)
        seed = gva.next_drop_seed()
        torch.manual_seed(99)  # the fused path draws the same seed again
        if mode == "unfused":
            attn.__dict__["_ao_drop_seed"] = seed
        out = attn(feat, coord, idx)
        grads = torch.autograd.grad(out, [feat] + list(attn.parameters()), gout)
        attn.__dict__.pop("_ao_drop_seed", None)
        res[mode] = (out.detach(), grads, {kk: v.clone() for kk, v in attn.state_dict().items()})
    of, gf, sf = res["fused"]
    ou, gu, su = res["unfused"]
    assert rel(of, ou) < 2e-5, rel(of, ou)
    names = ["feat"] + [nm for nm, _ in attn.named_parameters()]
    for nm, a, b in zip(names, gf, gu):
        assert _grad_close(nm, a, b), (nm, rel(a, b), float((a - b).abs().max()))
    for kk in su:
        np.testing.assert_allclose(sf[kk].cpu().numpy(), su[kk].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=kk)
    # and the mask matters: the same call without dropout differs at O(1)
    attn.eval()
    monkeypatch.setenv("AO_AMD_GVA", "fused")
    attn.load_state_dict(state0)
    with torch.no_grad():
        plain = attn(feat0, coord, idx)
    assert rel(plain, ou) > 0.05


def test_eval_mode_ignores_the_rate(monkeypatch):
    import ao_amd.ptv2 as ptv2

    torch.manual_seed(3)
    a = ptv2.GroupedVectorAttention(96, 12, attn_drop_rate=0.4).cuda()
    b = ptv2.GroupedVectorAttention(96, 12, attn_drop_rate=0.0).cuda()
    b.load_state_dict(a.state_dict())
    coord, idx = _neighbours(1500, 16, seed=5)
    feat = torch.randn(1500, 96, device="cuda")
    a.train()(feat, coord, idx)  # moves the running statistics; eval below uses them
    b.load_state_dict(a.state_dict())
    with torch.no_grad():
        ya, yb = a.eval()(feat, coord, idx), b.eval()(feat, coord, idx)
    assert torch.equal(ya, yb)


def test_fresh_mask_every_call_and_reproducible_under_manual_seed():
    import ao_amd.ptv2 as ptv2

    torch.manual_seed(4)
    attn = ptv2.GroupedVectorAttention(48, 6, attn_drop_rate=0.3).cuda().train()
    coord, idx = _neighbours(2000, 16, seed=6)
    feat = torch.randn(2000, 48, device="cuda")
    torch.manual_seed(11)
    y1, y2 = attn(feat, coord, idx).detach(), attn(feat, coord, idx).detach()
    torch.manual_seed(11)
    y3 = attn(feat, coord, idx).detach()
    assert not torch.equal(y1, y2)
    assert torch.equal(y1, y3)


def _block_modules(model):
    from ao_amd.ptv2 import native_model

    return native_model.runtime(model).block_modules


@pytest.mark.parametrize("tag,points,checkpoint", [("s3dis", 6000, False), ("scannet", 5000, False), ("s3dis", 6000, True)])
def test_native_model_with_attention_dropout(monkeypatch, tag, points, checkpoint):
    """Whole model, attn_drop_rate 0.2: the native runtime (one seed per Block per step, the backward -- and the checkpointed
    recomputation -- evaluates the forward's mask again) against the module-by-module python model whose attentions run the
    literal op sequence with the same masks."""
    import ao_amd.ptv2 as ptv2
    from ao_amd import synth
    from ao_amd.ptv2 import gva, native_model

    cfg = dict(M.S3DIS_CFG if tag == "s3dis" else M.SCANNET_CFG, drop_path_rate=0.0, attn_drop_rate=0.2,
               enable_checkpoint=checkpoint)
    b = synth.scene_batch([1, 2], point_max=points, in_channels=cfg["in_channels"], num_classes=cfg["num_classes"])
    data = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    res = {}
    for mode in ("native", "python"):
        monkeypatch.setenv("AO_AMD_MODEL", mode)
        if mode == "python":
            monkeypatch.setenv("AO_AMD_BLOCK", "python")
            monkeypatch.setenv("AO_AMD_GVA", "unfused")
        model = ptv2.PointTransformerV2(**cfg).cuda()
        model.load_state_dict(M.init_state(cfg, seed=17), strict=True)
        model.train()
        blocks = _block_modules(model)
        torch.manual_seed(5)
        seeds = [gva.next_drop_seed() for _ in blocks]
        torch.manual_seed(5)
        if mode == "native":
            assert native_model.supported(model, data["feat"])
        else:
            for blk, s in zip(blocks, seeds):
                blk.attn.__dict__["_ao_drop_seed"] = s
        logits = model(data)
        loss = F.cross_entropy(logits, data["segment"], ignore_index=-1)
        grads = torch.autograd.grad(loss, list(model.parameters()))
        res[mode] = (logits.detach(), grads, {k: v.clone() for k, v in model.state_dict().items()})
    ln, gn, sn = res["native"]
    lp, gp, sp = res["python"]
    np.testing.assert_allclose(ln.cpu().numpy(), lp.cpu().numpy(), rtol=0, atol=5e-5)
    names = [n for n, _ in model.named_parameters()]
    for nm, a, bb in zip(names, gn, gp):
        assert rel(a, bb) < 1e-2 or float((a - bb).abs().max()) < 2e-6, (nm, rel(a, bb), float((a - bb).abs().max()))
    for k in sp:
        np.testing.assert_allclose(sn[k].cpu().numpy(), sp[k].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    # dropout is on: a run without it is far away
    monkeypatch.setenv("AO_AMD_MODEL", "native")
    monkeypatch.delenv("AO_AMD_BLOCK", raising=False)
    monkeypatch.delenv("AO_AMD_GVA", raising=False)
    model = ptv2.PointTransformerV2(**dict(cfg, attn_drop_rate=0.0)).cuda()
    model.load_state_dict(M.init_state(cfg, seed=17), strict=True)
    plain = model.train()(data).detach()
    assert rel(plain, lp) > 1e-2


def test_native_block_with_attention_dropout(monkeypatch):
    """One Block through csrc/block.hip (AO_AMD_MODEL=python keeps the per-Block runtime) against the python Block."""
    import ao_amd.ptv2 as ptv2
    from ao_amd.ptv2 import gva

    torch.manual_seed(8)
    blk = ptv2.Block(96, 12, attn_drop_rate=0.25).cuda().train()
    coord, idx = _neighbours(4000, 16, seed=9)
    offset = torch.tensor([2000, 4000], dtype=torch.int32).cuda()
    feat0 = torch.randn(4000, 96, device="cuda")
    gout = torch.randn(4000, 96, device="cuda")
    state0 = {k: v.clone() for k, v in blk.state_dict().items()}
    res = {}
    for mode in ("native", "python"):
        if mode == "python":
            monkeypatch.setenv("AO_AMD_BLOCK", "python")
            monkeypatch.setenv("AO_AMD_GVA", "unfused")
        blk.load_state_dict(state0)
        torch.manual_seed(21)
        seed = gva.next_drop_seed()
        torch.manual_seed(21)
        if mode == "python":
            blk.attn.__dict__["_ao_drop_seed"] = seed
        feat = feat0.clone().requires_grad_(True)
        out = blk([coord, feat, offset], idx)[1]
        grads = torch.autograd.grad(out, [feat] + list(blk.parameters()), gout)
        res[mode] = (out.detach(), grads)
    assert rel(res["native"][0], res["python"][0]) < 2e-5
    names = ["feat"] + [nm for nm, _ in blk.named_parameters()]
    for nm, a, b in zip(names, res["native"][1], res["python"][1]):
        assert _grad_close(nm, a, b), (nm, rel(a, b), float((a - b).abs().max()))


def test_graph_issue_carries_the_seed(monkeypatch):
    """The seed is a kernel argument: a step issued as an updated graph uses this step's seeds, not the captured ones."""
    import ao_amd.ptv2 as ptv2
    from ao_amd import synth

    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0, attn_drop_rate=0.2)
    b = synth.scene_batch([3], point_max=5000, in_channels=cfg["in_channels"], num_classes=cfg["num_classes"])
    data = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    from ao_amd import _lib

    L = _lib.lib()
    prev = L.ptv2_graph_mode(-1)
    outs = {}
    for graph in ("1", "0"):
        L.ptv2_graph_mode(int(graph))
        model = ptv2.PointTransformerV2(**cfg).cuda()
        model.load_state_dict(M.init_state(cfg, seed=3), strict=True)
        model.train()
        torch.manual_seed(77)
        steps = []
        for _ in range(4):
            logits = model(data)
            g = torch.autograd.grad(logits.square().mean(), list(model.parameters()))
            steps.append((logits.detach().clone(), [x.clone() for x in g]))
        outs[graph] = steps
    L.ptv2_graph_mode(prev)
    L.ptv2_graph_reset()
    for (la, ga), (lb, gb) in zip(outs["1"], outs["0"]):
        assert torch.equal(la, lb)
        for x, y in zip(ga, gb):
            assert torch.equal(x, y)
    assert not torch.equal(outs["1"][0][0], outs["1"][1][0])
