"""GPU: the whole-model native runtime (ao_amd/csrc/model.hip: PointTransformerV2.forward / backward as one call per
direction, point_transformer_v2m2_base.py:556-576) against the same module evaluated stage by stage in python
(AO_AMD_MODEL=python: one native call per Block, autograd nodes for GridPool / Unpool / head) -- which the fixture tests
of tests/test_gpu_model.py pin to the reference nn.Module.  Logits 1e-5 (same kernels, same order except the unpool's
add), every parameter gradient in relative L2; the two ways of delivering parameter gradients; the optimizer's zero-copy
path; eval mode; DropPath."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ptv2_ref as M

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _data(seeds, points, cfg):
    from ao_amd import synth

    b = synth.scene_batch(seeds, point_max=points, in_channels=cfg["in_channels"], num_classes=cfg["num_classes"])
    return {k: torch.from_numpy(v).cuda() for k, v in b.items()}


def _model(cfg, seed, train=True):
    import ao_amd.ptv2 as ptv2

    m = ptv2.PointTransformerV2(**cfg).cuda()
    m.load_state_dict(M.init_state(cfg, seed=seed), strict=True)
    return m.train(train)


@pytest.mark.parametrize("tag,points", [("s3dis", 6000), ("scannet", 5000), ("s3dis", 40000)])
def test_native_model_matches_stagewise_python(monkeypatch, tag, points):
    from ao_amd.ptv2 import native_model

    cfg = dict(M.S3DIS_CFG if tag == "s3dis" else M.SCANNET_CFG, drop_path_rate=0.0)
    data = _data([1, 2], points, cfg)
    res = {}
    for mode in ("native", "python"):
        monkeypatch.setenv("AO_AMD_MODEL", mode)
        model = _model(cfg, seed=17)
        assert native_model.supported(model, data["feat"]) == (mode == "native")
        logits = model(data)
        loss = F.cross_entropy(logits, data["segment"], ignore_index=-1)
        grads = torch.autograd.grad(loss, list(model.parameters()))
        res[mode] = (logits.detach(), grads, {k: v.clone() for k, v in model.state_dict().items()})
    ln, gn, sn = res["native"]
    lp, gp, sp = res["python"]
    np.testing.assert_allclose(ln.cpu().numpy(), lp.cpu().numpy(), rtol=0, atol=2e-5)
    names = [n for n, _ in model.named_parameters()]
    for nm, a, b in zip(names, gn, gp):
        assert a.shape == b.shape
        # (two fp32 evaluations of the same network whose Linear + BatchNorm layers outside the Blocks sum in different orders:
        # the earliest layer's gradient -- patch_embed.proj, behind all 15 Blocks' ReLU masks -- has been observed between
        # 1.2e-3 and 2.1e-3 over this repository's builds, and a single pre-activation whose sign differs between the two
        # evaluations moves ONE element of a small BatchNorm gradient by a few percent (5.6e-3 of the vector's norm seen once;
        # tools/gpu/diag_outlier.py: deterministic per build, a different element / none at all for other seeds).  The
        # bound is what separates that from a wiring error, which shows up at O(1))
        assert rel(a, b) < 1e-2 or float((a - b).abs().max()) < 2e-6, (nm, rel(a, b), float((a - b).abs().max()))
    for k in sp:  # BatchNorm running statistics / batch counters after one training forward
        np.testing.assert_allclose(sn[k].cpu().numpy(), sp[k].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


def test_native_model_eval_and_no_grad(monkeypatch):
    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.3)
    data = _data([3], 8000, cfg)
    out = {}
    for mode in ("native", "python"):
        monkeypatch.setenv("AO_AMD_MODEL", mode)
        model = _model(cfg, seed=5, train=False)
        with torch.no_grad():
            out[mode] = model(data)
    np.testing.assert_allclose(out["native"].cpu().numpy(), out["python"].cpu().numpy(), rtol=0, atol=2e-5)


def test_direct_parameter_gradients_and_zero_copy_optimizer():
    """`native_param_grads = "direct"`: the node assigns `.grad` itself (no AccumulateGrad), gradients alias one flat
    buffer in FlatAdamW's layout, the optimizer consumes it without a copy; a second backward without zero_grad adds."""
    from ao_amd.ptv2.optim import FlatAdamW

    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    data = _data([4], 5000, cfg)
    ref = _model(cfg, seed=9)
    loss = F.cross_entropy(ref(data), data["segment"], ignore_index=-1)
    loss.backward()
    want = [p.grad.clone() for p in ref.parameters()]

    model = _model(cfg, seed=9)
    model.native_param_grads = "direct"
    opt = FlatAdamW(model.parameters(), lr=0.006, weight_decay=0.05)
    F.cross_entropy(model(data), data["segment"], ignore_index=-1).backward()
    got = [p.grad for p in model.parameters()]
    for a, b in zip(got, want):
        assert torch.equal(a, b)  # the same kernels wrote both
    flat = opt.flatten_grads()
    assert flat.data_ptr() == got[0].data_ptr() and flat.numel() == opt._n  # zero copy: the buffer itself
    assert flat.data_ptr() != opt.flat_grad.data_ptr()
    # accumulation: a second backward onto existing gradients adds
    model.load_state_dict(M.init_state(cfg, seed=9), strict=True)  # undo the running-statistics update
    F.cross_entropy(model(data), data["segment"], ignore_index=-1).backward()
    for p, b in zip(model.parameters(), want):
        assert rel(p.grad, 2 * b) < 1e-6
    # after zero_grad the optimizer step equals torch.optim.AdamW's on the same gradients
    opt.zero_grad(set_to_none=True)
    F.cross_entropy(model(data), data["segment"], ignore_index=-1).backward()
    before = [p.detach().clone() for p in model.parameters()]
    grads = [p.grad.clone() for p in model.parameters()]
    opt.step(flat_grad=opt.flatten_grads())
    twin = [torch.nn.Parameter(b.clone()) for b in before]
    topt = torch.optim.AdamW(twin, lr=0.006, weight_decay=0.05)
    for t, g in zip(twin, grads):
        t.grad = g
    topt.step()
    for p, t in zip(model.parameters(), twin):
        np.testing.assert_allclose(p.detach().cpu().numpy(), t.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)


def test_droppath_rows_pass_the_identity_through():
    """DropPath is per point (timm DropPath on (N,C), :160-162): with rate r a fraction ~r of a block's rows keeps its
    input; the native runtime draws all blocks' factors in one call -- statistics and determinism under a seed."""
    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.3)
    data = _data([6], 6000, cfg)
    model = _model(cfg, seed=3)
    torch.manual_seed(123)
    a = model(data).detach()
    model.load_state_dict(M.init_state(cfg, seed=3), strict=True)
    torch.manual_seed(123)
    b = model(data).detach()
    assert torch.equal(a, b)
    model.load_state_dict(M.init_state(cfg, seed=3), strict=True)
    torch.manual_seed(124)
    c = model(data).detach()
    assert not torch.equal(a, c)
    from ao_amd.ptv2 import native_model

    rt = native_model.runtime(model)
    geo = model.geometry(data["coord"], data["offset"].int())
    scales = rt.draw_droppath(geo, data["coord"].device)
    rates = [r for seq in rt.droppath for r in seq if r > 0]
    assert len(rates) == 11 and abs(max(rates) - 0.3) < 1e-6  # linspace(0, 0.3, 10) encoder + linspace(0, 0.3, 3) decoder, zeros skipped
    zero_frac = float((scales == 0).float().mean())
    assert 0.05 < zero_frac < 0.3
    assert torch.all((scales == 0) | (scales > 1.0))


def test_training_loop_with_changing_scenes_matches_the_python_path(monkeypatch):
    """Real training sees a different scene (different N at every level) each step: the cached runtime struct, the
    DropPath layout cache, the persistent gradient buffer ("direct" mode) and the geometry prefetcher must all follow.
    Six optimizer steps alternating three scenes, native runtime + direct gradients + prefetch against the stage-by-stage
    python path with autograd-delivered gradients: same loss at every step (drop_path 0; 1e-4 absolute, Adam-amplified
    summation-order noise: the first two steps agree to 1e-5, later ones to ~1e-3), parameters within 2e-2 relative L2 at the end."""
    import ao_amd.ptv2 as ptv2
    from ao_amd.ptv2 import parallel
    from ao_amd.ptv2.optim import FlatAdamW

    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    scenes = [_data([s], n, cfg) for s, n in ((1, 9000), (2, 5000), (3, 12000))]
    for d in scenes:  # learnable labels
        d["segment"] = (d["coord"][:, 2] * 3).long().clamp(0, 12)
    curves, finals = {}, {}
    for mode in ("native", "python"):
        monkeypatch.setenv("AO_AMD_MODEL", mode)
        torch.manual_seed(0)
        seg = ptv2.DefaultSegmentor(cfg).cuda().train()
        seg.backbone.load_state_dict(M.init_state(cfg, seed=11), strict=True)
        if mode == "native":
            seg.backbone.native_param_grads = "direct"
        opt = FlatAdamW(seg.parameters(), lr=0.003, weight_decay=0.05)
        pre = parallel.GeometryPrefetcher(seg.backbone, torch.device("cuda"))
        pre.start(scenes[0]["coord"], scenes[0]["offset"])
        losses = []
        for step in range(6):
            data = scenes[step % 3]
            loss = seg(dict(data, geometry=pre.take()))["loss"]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            nxt = scenes[(step + 1) % 3]
            pre.start(nxt["coord"], nxt["offset"])
            opt.step(flat_grad=opt.flatten_grads())
            losses.append(float(loss.detach()))
        pre.take()
        curves[mode] = losses
        finals[mode] = [p.detach().clone() for p in seg.parameters()]
    print("native", ["%.5f" % v for v in curves["native"]], "\npython", ["%.5f" % v for v in curves["python"]])
    assert curves["native"][-1] < curves["native"][0]
    # (steps 3-6 carry Adam-amplified summation-order noise: 1e-3 ... 5e-3 over this round's builds; the first two steps are
    # pinned tightly below)
    # What this test pins is WIRING (a stale table, a wrong level size or a missed prefetch shows up at O(1) from the step it
    # happens in), not arithmetic: the first step -- same weights, same scene -- agrees to 1e-5.  From the second step on the
    # two trajectories separate chaotically: Adam's first update is lr * sign(g) for every parameter, including the biases
    # whose true gradient is exactly zero and whose computed gradient is summation-order noise, and 12 000-point scenes put a
    # few dozen points into the deepest level, where one flipped ReLU moves whole BatchNorm statistics.  Observed over this
    # repository's builds (each changing some fp32 summation order): step 2 1e-5 ... 5e-4, steps 3-6 1e-3 ... 4e-2, final
    # parameters 7e-3 ... 2e-2 relative.
    np.testing.assert_allclose(curves["native"], curves["python"], rtol=0, atol=8e-2)
    assert abs(curves["native"][0] - curves["python"][0]) < 1e-5 and abs(curves["native"][1] - curves["python"][1]) < 2e-3
    num = sum(float((a.double() - b.double()).pow(2).sum()) for a, b in zip(finals["native"], finals["python"]))
    den = sum(float(b.double().pow(2).sum()) for b in finals["python"])
    assert (num / den) ** 0.5 < 5e-2, (num / den) ** 0.5


@pytest.mark.parametrize("variant", ["two_stages_no_qkv_bias", "five_stages_map"])
def test_native_model_on_other_architectures(monkeypatch, variant):
    """The runtime description is generic in the number of stages, depths, widths, neighbour counts, qkv bias and unpool
    backend (ptv2_model: up to 5 stages / 40 blocks): two off-config networks, native against stage-by-stage python."""
    import ao_amd.ptv2 as ptv2
    from ao_amd import synth
    from ao_amd.ptv2 import native_model

    if variant == "two_stages_no_qkv_bias":
        cfg = dict(in_channels=6, num_classes=7, patch_embed_depth=1, patch_embed_channels=48, patch_embed_groups=6,
                   patch_embed_neighbours=8, enc_depths=(1, 3), enc_channels=(96, 192), enc_groups=(12, 24),
                   enc_neighbours=(16, 8), dec_depths=(2, 1), dec_channels=(48, 96), dec_groups=(6, 12), dec_neighbours=(8, 16),
                   grid_sizes=(0.12, 0.3), attn_qkv_bias=False, pe_multiplier=False, pe_bias=True, attn_drop_rate=0.0,
                   drop_path_rate=0.0, enable_checkpoint=False, unpool_backend="interp")
    else:
        cfg = dict(in_channels=9, num_classes=20, patch_embed_depth=1, patch_embed_channels=48, patch_embed_groups=6,
                   patch_embed_neighbours=16, enc_depths=(1, 1, 1, 1, 1), enc_channels=(48, 96, 96, 192, 384),
                   enc_groups=(6, 12, 12, 24, 48), enc_neighbours=(16, 16, 16, 16, 16), dec_depths=(1, 1, 1, 1, 1),
                   dec_channels=(48, 48, 96, 96, 192), dec_groups=(6, 6, 12, 12, 24), dec_neighbours=(16, 16, 16, 16, 16),
                   grid_sizes=(0.08, 0.16, 0.3, 0.6, 1.2), attn_qkv_bias=True, pe_multiplier=False, pe_bias=True,
                   attn_drop_rate=0.0, drop_path_rate=0.0, enable_checkpoint=False, unpool_backend="map")
    b = synth.scene_batch([1, 2], point_max=9000, in_channels=cfg["in_channels"], num_classes=cfg["num_classes"])
    data = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    res = {}
    for mode in ("native", "python"):
        monkeypatch.setenv("AO_AMD_MODEL", mode)
        torch.manual_seed(5)
        model = ptv2.PointTransformerV2(**cfg).cuda().train()
        assert native_model.supported(model, data["feat"]) == (mode == "native")
        logits = model(data)
        loss = F.cross_entropy(logits, data["segment"], ignore_index=-1)
        res[mode] = (logits.detach(), torch.autograd.grad(loss, list(model.parameters())))
    np.testing.assert_allclose(res["native"][0].cpu().numpy(), res["python"][0].cpu().numpy(), rtol=0, atol=2e-5)
    for (nm, _), a, b_ in zip(model.named_parameters(), res["native"][1], res["python"][1]):
        # (2e-2, the gradient bound of the 120 k oracle test: both sides are fp32 HIP paths with different summation orders -- the
        # native model takes the BatchNorm statistics of the layers between the Blocks from the GEMM's 64-row records, the q / k
        # BatchNorm backward sums from the launch that forms their gradients -- and a 1e-6 difference in a pre-activation flips a
        # ReLU mask, which moves a whole term of the small gradients at the end of the backward chain)
        assert rel(a, b_) < 2e-2 or float((a - b_).abs().max()) < 2e-6, (nm, rel(a, b_))


def test_native_activation_checkpointing_gives_the_same_bits_with_less_memory():
    """enable_checkpoint=True (reference :169-171) on the native runtime (ptv2_model.checkpoint): the forward keeps only the
    Blocks' outputs, the backward re-runs each Block's forward into ONE shared region right before its backward.  Same
    kernels on the same inputs: logits and every gradient equal the non-checkpointed run bit for bit; the saved arena shrinks.
    Running statistics as under torch.utils.checkpoint in the reference: the norms inside the checkpointed attention take the
    momentum step twice per training step (and count two batches), norm1 / norm2 / norm3 once."""
    import ctypes

    from ao_amd import _lib
    from ao_amd.ptv2 import native_model

    data = _data([5], 9000, dict(M.S3DIS_CFG))
    res, saved = {}, {}
    for ck in (False, True):
        cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0, enable_checkpoint=ck)
        model = _model(cfg, seed=23)
        assert native_model.supported(model, data["feat"])
        logits = model(data)
        loss = F.cross_entropy(logits, data["segment"], ignore_index=-1)
        grads = torch.autograd.grad(loss, list(model.parameters()))
        res[ck] = (logits.detach(), grads, {k: v.clone() for k, v in model.state_dict().items()})
        rt = native_model.runtime(model)
        assert int(rt.M.checkpoint) == int(ck)
        saved[ck] = int(_lib.lib().ptv2_model_saved_bytes(ctypes.addressof(rt.M)))
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)
    init = M.init_state(cfg, seed=23)
    momentum, twice = 0.1, 0  # (nn.BatchNorm1d's default, what PointBatchNorm constructs)
    for k in res[False][2]:
        once, ck = res[False][2][k], res[True][2][k]
        inside = ".attn." in k and ("running_" in k or "num_batches_tracked" in k)
        if not inside:
            assert torch.equal(ck, once), k
        elif "num_batches_tracked" in k:
            assert int(ck) == int(once) + 1 == 2, k
        else:  # r2 = (1 - m) r1 + m b with m b = r1 - (1 - m) r0
            r0 = torch.as_tensor(init[k]).cuda().float()
            expect = (1 - momentum) * once + (once - (1 - momentum) * r0)
            np.testing.assert_allclose(ck.cpu().numpy(), expect.cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
            twice += 1
    assert twice == 2 * 4 * 15  # mean and variance of four norms in each of the 15 Blocks
    # (the deep levels no longer save an (N,G,C) tensor: the full arena shrank, the shared region is still the level-0 Block's)
    assert saved[True] < 0.8 * saved[False], saved


@pytest.mark.parametrize("tag,points,bf16", [("s3dis", 9000, False), ("scannet", 7000, False), ("s3dis", 9000, True)])
def test_deferred_weight_gradients_equal_the_per_block_launches(tag, points, bf16):
    """ptv2_wgrad_defer_mode: the weight gradients filed and run by a few batched launches at the end of the backward (default)
    against every launch where it is called, over two steps with different scenes (the job table is rebuilt per call) and with
    drop_path on.  Same kernels and tile order; a filed job sums over longer row chunks (fewer records), so a weight gradient
    may differ in its last bits: logits and every other gradient bit for bit, the weight gradients to 2e-6 of their norm."""
    from ao_amd import _lib

    L = _lib.lib()
    cfg = dict(M.S3DIS_CFG if tag == "s3dis" else M.SCANNET_CFG, drop_path_rate=0.3)
    prev = L.ptv2_wgrad_defer_mode(-1)
    res = {}
    try:
        for mode in (1, 0):
            L.ptv2_wgrad_defer_mode(mode)
            model = _model(cfg, seed=31)
            torch.manual_seed(7)
            steps = []
            for seeds, pts in (([3, 4], points), ([5], points + 1500)):
                data = _data(seeds, pts, cfg)
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):  # (bf16: the direct-form kernels' jobs)
                    logits = model(data)
                    loss = F.cross_entropy(logits.float(), data["segment"], ignore_index=-1)
                grads = torch.autograd.grad(loss, list(model.parameters()))
                steps.append((logits.detach().clone(), [g.clone() for g in grads]))
            res[mode] = steps
    finally:
        L.ptv2_wgrad_defer_mode(prev)
    names = [n for n, _ in model.named_parameters()]
    for (la, ga), (lb, gb) in zip(res[1], res[0]):
        assert torch.equal(la, lb)
        differ = 0
        for nm, a, b in zip(names, ga, gb):
            if not torch.equal(a, b):
                differ += 1
                # (a bias in front of a training-mode BatchNorm has an exactly zero gradient: both sides hold summation noise)
                assert rel(a, b) < (2e-3 if bf16 else 2e-6) or float((a - b).abs().max()) < 1e-8, (nm, rel(a, b), float((a - b).abs().max()))
        assert differ < len(names) // 2  # (BatchNorm parameters, input-side gradients, the attention's folded parameters: untouched)
