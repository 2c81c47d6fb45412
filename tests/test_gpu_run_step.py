"""GPU: the reference trainer's `run_step` AS WRITTEN (pointcept/engines/train_sam_pp2s.py:178-196, enable_amp=True) over the
native runtime: `torch.cuda.amp.autocast()` with its default dtype, `GradScaler.scale(loss).backward()`,
`scaler.step(optimizer)`, `scaler.update()`, and the scheduler stepped only when the scale did not back off -- with
`FlatAdamW` as the optimizer and the native backward delivering `.grad` as views of one flat buffer
(`native_param_grads="direct"`): `unscale_` runs in place on those views, the inf check sees them, a skipped step leaves
the weights bit-identical.  "AO training loop unchanged" (BASELINE.json north_star) means exactly this statement
sequence runs; nothing in the trainer is edited (no `enabled=False`).

Arithmetic under that autocast: the reference's Linear layers run in fp16 (CUDA default of torch.cuda.amp.autocast); here
they run on the bf16 matrix cores with fp32 storage (ao_amd/ptv2/native_model.py: matmul_bf16) -- wider exponent, so the
scaler never has to back off on its own; its back-off path is exercised by injecting an inf."""
import warnings

import numpy as np
import pytest
import torch

from oracle import ptv2_ref as M

pytestmark = pytest.mark.gpu

STEPS, OVERFLOW_AT = 10, 4


def _batch(cfg):
    from ao_amd import synth

    b = synth.scene_batch([31, 32], point_max=6000, in_channels=cfg["in_channels"], num_classes=cfg["num_classes"])
    return {k: torch.from_numpy(v) for k, v in b.items()}  # host tensors: run_step moves them (:175-177)


def _trainer(cfg, flat):
    import ao_amd.ptv2 as ptv2
    from ao_amd.ptv2.schedule import build_optimizer, build_scheduler

    model = ptv2.DefaultSegmentor(dict(cfg)).cuda().train()
    model.backbone.load_state_dict(M.init_state(cfg, seed=23), strict=True)
    # configs/s3dis/semseg-pt-v2m2-0-base.py:41-43
    optimizer = build_optimizer(dict(type="AdamW", lr=0.006, weight_decay=0.05), model, flat=flat)
    scheduler = build_scheduler(dict(type="MultiStepLR", milestones=[0.6, 0.8], gamma=0.1), optimizer, total_steps=STEPS)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (torch.cuda.amp.GradScaler is the spelling the reference uses; deprecated alias)
        scaler = torch.cuda.amp.GradScaler()
    return model, optimizer, scheduler, scaler


def _run_step(model, optimizer, scheduler, scaler, input_dict, poison=None):
    """The statements of train_sam_pp2s.py:173-196 (enable_amp branch), in their order."""
    for key in input_dict.keys():
        if isinstance(input_dict[key], torch.Tensor):
            input_dict[key] = input_dict[key].cuda(non_blocking=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.cuda.amp.autocast(enabled=True):
            output_dict = model(input_dict)
            loss = output_dict["loss"]
    optimizer.zero_grad()
    scaler.scale(loss).backward()
    if poison is not None:  # (test only: what an fp16 overflow leaves in a gradient)
        poison()
    scaler.step(optimizer)
    scale = scaler.get_scale()
    scaler.update()
    stepped = scale <= scaler.get_scale()
    if stepped:
        scheduler.step()
    return float(loss.detach()), stepped


def _train(monkeypatch, mode):
    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    monkeypatch.setenv("AO_AMD_MODEL", mode)
    flat = mode == "native"
    model, optimizer, scheduler, scaler = _trainer(cfg, flat)
    if flat:
        model.backbone.native_param_grads = "direct"
    batch = _batch(cfg)
    log = []
    for it in range(STEPS):
        before = [p.detach().clone() for p in model.parameters()]
        scale0 = scaler.get_scale()
        poison = None
        if it == OVERFLOW_AT:
            victim = list(model.parameters())[5]

            def poison(victim=victim):
                victim.grad.view(-1)[0] = float("inf")
        loss, stepped = _run_step(model, optimizer, scheduler, scaler, dict(batch), poison)
        after = [p.detach() for p in model.parameters()]
        changed = any(not torch.equal(a, b) for a, b in zip(before, after))
        log.append(dict(loss=loss, stepped=stepped, changed=changed, scale0=scale0, scale1=scaler.get_scale(),
                        lr=optimizer.param_groups[0]["lr"]))
    return model, optimizer, log


def test_reference_run_step_statements_over_the_native_runtime(monkeypatch):
    from ao_amd.ptv2.optim import FlatAdamW

    model, optimizer, log = _train(monkeypatch, "native")
    assert isinstance(optimizer, FlatAdamW)
    # gradients were views of ONE flat buffer (the zero-copy path), unscaled in place by the scaler
    params = list(model.parameters())
    flat = optimizer.flatten_grads()
    assert flat.data_ptr() == params[0].grad.data_ptr() and flat.data_ptr() != optimizer.flat_grad.data_ptr()
    for it, rec in enumerate(log):
        assert np.isfinite(rec["loss"])
        if it == OVERFLOW_AT:
            # the forced overflow: optimizer.step skipped -> weights bit-identical; the scale backs off; scheduler not stepped
            assert not rec["changed"] and not rec["stepped"] and rec["scale1"] == 0.5 * rec["scale0"], rec
        else:
            assert rec["changed"] and rec["stepped"] and rec["scale1"] == rec["scale0"], (it, rec)
    # MultiStepLR(0.6, 0.8 of 10 steps), stepped 9 times (one skip): milestones at scheduler steps 6 and 8
    lrs = [rec["lr"] for rec in log]
    np.testing.assert_allclose(lrs, [0.006] * 6 + [0.0006] * 2 + [0.00006] * 2, rtol=1e-6)
    assert optimizer._step == STEPS - 1
    assert log[-1]["loss"] < log[0]["loss"]  # it trains (same batch every step)

    # the same statements over the stage-by-stage python path with torch.optim.AdamW: the trajectories agree.  (AdamW
    # normalises every gradient element by its own running magnitude, so after 10 steps of lr 0.006 an element whose
    # gradient is noise-level has moved +-0.06 either way in BOTH runs: the weights decorrelate at that scale -- 0.14 of
    # their norm observed -- while the loss curves stay together; the gradients themselves are pinned below.)
    model_p, optimizer_p, log_p = _train(monkeypatch, "python")
    assert isinstance(optimizer_p, torch.optim.AdamW)
    assert [r["stepped"] for r in log_p] == [r["stepped"] for r in log]
    assert abs(log[0]["loss"] - log_p[0]["loss"]) < 2e-3  # one forward, bf16 products in both
    np.testing.assert_allclose([r["loss"] for r in log], [r["loss"] for r in log_p], rtol=0, atol=8e-2)
    num = sum(float((a.detach().double() - b.detach().double()).pow(2).sum()) for a, b in zip(model.parameters(), model_p.parameters()))
    den = sum(float(b.detach().double().pow(2).sum()) for b in model_p.parameters())
    assert (num / den) ** 0.5 < 0.3, (num / den) ** 0.5


def test_gradients_after_unscale_are_the_unscaled_gradients(monkeypatch):
    """What optimizer.step sees inside scaler.step: `scale(loss).backward()` then `unscale_` in place on the flat buffer's
    views must give the gradients of `loss.backward()` -- the scale is a power of two and nothing overflows in fp32 storage,
    so bit for bit -- for every parameter tensor (a mis-wired slot or a missed unscale would not show in AdamW's
    scale-invariant update)."""
    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    monkeypatch.setenv("AO_AMD_MODEL", "native")
    got = {}
    for use_scaler in (True, False):
        model, optimizer, scheduler, scaler = _trainer(cfg, True)
        model.backbone.native_param_grads = "direct"
        batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in _batch(cfg).items()}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with torch.cuda.amp.autocast(enabled=True):
                loss = model(batch)["loss"]
        optimizer.zero_grad()
        if use_scaler:
            scaler.scale(loss).backward()
            scaler.unscale_(optimizer)
        else:
            loss.backward()
        got[use_scaler] = [p.grad.detach().clone() for p in model.parameters()]
    assert len(got[True]) == len(got[False]) > 400
    for a, b in zip(got[True], got[False]):
        assert torch.equal(a, b)


def test_run_step_statements_against_the_cpu_oracle(monkeypatch):
    """ONE oracle-side run of the same statement sequence.  The CPU oracle (oracle/ptv2_ref.RefModule, the
    reference network restated on CPU torch) under `torch.optim.AdamW` + torch's own MultiStepLR, and the HIP path under
    FlatAdamW + StepSchedule, both driven by train_sam_pp2s.py:173-200 with enable_amp=False (a GradScaler on CPU tensors is a
    no-op, and fp32 on both sides is the comparison that pins arithmetic): same loss at step 0, the loss curves together over
    10 optimizer steps of lr 0.006, the same learning-rate sequence, every step taken."""
    import torch.nn.functional as F

    import ao_amd.ptv2 as ptv2
    from ao_amd.ptv2.schedule import build_optimizer, build_scheduler

    monkeypatch.setenv("AO_AMD_MODEL", "native")
    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    host = _batch(cfg)
    # labels that are a function of position: something to learn (random labels give a flat curve)
    z = host["coord"][:, 2]
    host["segment"] = ((z - z.min()) / (z.max() - z.min() + 1e-6) * 12.99).long()

    class OracleSegmentor(torch.nn.Module):  # DefaultSegmentor (pointcept/models/default.py:232-251) over the oracle backbone
        def __init__(self):
            super().__init__()
            self.backbone = M.RefModule(cfg, seed=23, randomize_bn=False)

        def forward(self, input_dict):
            return dict(loss=F.cross_entropy(self.backbone(input_dict), input_dict["segment"], ignore_index=-1))

    def run(model, optimizer, scheduler, move):
        log = []
        for _ in range(STEPS):
            input_dict = dict(host)
            for key in input_dict.keys():  # :175-177
                if isinstance(input_dict[key], torch.Tensor):
                    input_dict[key] = move(input_dict[key])
            output_dict = model(input_dict)  # enable_amp False: :178-180 without the autocast
            loss = output_dict["loss"]
            optimizer.zero_grad()
            loss.backward()      # :193
            optimizer.step()     # :194
            scheduler.step()     # :195
            log.append((float(loss.detach()), optimizer.param_groups[0]["lr"]))
        return log

    oracle = OracleSegmentor().train()
    opt_o = torch.optim.AdamW(oracle.parameters(), lr=0.006, weight_decay=0.05)
    sch_o = torch.optim.lr_scheduler.MultiStepLR(opt_o, milestones=[6, 8], gamma=0.1)  # (0.6, 0.8) of 10 steps
    log_o = run(oracle, opt_o, sch_o, lambda t: t)

    model = ptv2.DefaultSegmentor(dict(cfg)).cuda().train()
    model.backbone.load_state_dict(M.init_state(cfg, seed=23, randomize_bn=False), strict=True)
    opt = build_optimizer(dict(type="AdamW", lr=0.006, weight_decay=0.05), model, flat=True)
    sch = build_scheduler(dict(type="MultiStepLR", milestones=[0.6, 0.8], gamma=0.1), opt, total_steps=STEPS)
    log_h = run(model, opt, sch, lambda t: t.cuda(non_blocking=True))

    lo, lh = np.asarray([l for l, _ in log_o]), np.asarray([l for l, _ in log_h])
    print("loss oracle", np.round(lo, 5), "\nloss hip   ", np.round(lh, 5))
    assert abs(lo[0] - lh[0]) < 2e-5                      # the same forward
    np.testing.assert_allclose([r for _, r in log_h], [r for _, r in log_o], rtol=1e-6)  # the same schedule
    assert lh[-1] < 0.8 * lh[0] and lo[-1] < 0.8 * lo[0]  # both learn
    assert np.max(np.abs(lh - lo) / lo) < 0.06, (lo, lh)  # the curves stay together (trajectory test: 2.4 % observed)
