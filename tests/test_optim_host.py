"""Host logic of ao_amd/ptv2/optim.FlatAdamW and schedule.StepSchedule checkpoints on CPU tensors (the update kernel
is replaced by tests/_cpu_adamw.py's torch statement): the torch.optim.AdamW checkpoint layout the reference's
CheckpointSaver / CheckpointLoader write and read (pointcept/engines/hooks/misc.py:180-184, 247-248)."""
import copy

import pytest
import torch

from ao_amd.ptv2.optim import FlatAdamW
from ao_amd.ptv2.schedule import StepSchedule
from tests._cpu_adamw import TorchStatementAdamW


def _model(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.LayerNorm(5), torch.nn.Linear(5, 3))


def _grads(model, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(16, 7, generator=g)
    model.zero_grad(set_to_none=True)
    model(x).square().mean().backward()


def _same(a, b, tol=0.0):
    for (ka, pa), (kb, pb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb
        assert torch.allclose(pa, pb, rtol=0, atol=tol), (ka, float((pa - pb).abs().max()))


def test_cpu_step_without_kernel_raises():
    m = _model()
    opt = FlatAdamW(m.parameters(), lr=0.01)
    _grads(m, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


def test_flat_views_and_update_follow_torch_adamw():
    a, b = _model(), _model()
    oa = TorchStatementAdamW(a.parameters(), lr=0.006, weight_decay=0.05)
    ob = torch.optim.AdamW(b.parameters(), lr=0.006, weight_decay=0.05)
    for s in range(5):
        _grads(a, s), _grads(b, s)
        oa.step(), ob.step()
    _same(a, b, 1e-6)
    # parameters are views of the flat buffer
    assert all(p.data_ptr() >= oa.flat_param.data_ptr() for p in a.parameters())


def test_state_dict_is_torch_adamw_layout_and_round_trips():
    a, b = _model(), _model()
    oa = TorchStatementAdamW(a.parameters(), lr=0.006, weight_decay=0.05)
    ob = torch.optim.AdamW(b.parameters(), lr=0.006, weight_decay=0.05)
    assert oa.state_dict()["state"] == {}  # like torch before the first step
    for s in range(3):
        _grads(a, s), _grads(b, s)
        oa.step(), ob.step()
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sorted(sa["state"]) == sorted(sb["state"]) == list(range(6))
    for i in sb["state"]:
        assert set(sa["state"][i]) == {"step", "exp_avg", "exp_avg_sq"}
        assert float(sa["state"][i]["step"]) == float(sb["state"][i]["step"]) == 3.0
        assert sa["state"][i]["exp_avg"].shape == sb["state"][i]["exp_avg"].shape
        assert torch.allclose(sa["state"][i]["exp_avg"], sb["state"][i]["exp_avg"], atol=1e-7)
        assert torch.allclose(sa["state"][i]["exp_avg_sq"], sb["state"][i]["exp_avg_sq"], atol=1e-9)
    assert sa["param_groups"][0]["params"] == sb["param_groups"][0]["params"]

    # save -> fresh process -> reload -> continue == uninterrupted run (flat -> flat, torch -> flat, flat -> torch)
    ckpt_model, ckpt_flat, ckpt_torch = copy.deepcopy(a.state_dict()), copy.deepcopy(sa), copy.deepcopy(sb)
    for s in range(3, 6):
        _grads(a, s)
        oa.step()
    for source in (ckpt_flat, ckpt_torch):
        c = _model(seed=9)
        c.load_state_dict(ckpt_model)
        oc = TorchStatementAdamW(c.parameters(), lr=1.0)  # lr comes back from the checkpoint
        oc.load_state_dict(source)
        assert oc.param_groups[0]["lr"] == 0.006 and oc._step == 3
        for s in range(3, 6):
            _grads(c, s)
            oc.step()
        _same(a, c, 1e-6)
    d = _model(seed=9)
    d.load_state_dict(ckpt_model)
    od = torch.optim.AdamW(d.parameters(), lr=0.006, weight_decay=0.05)
    od.load_state_dict(ckpt_flat)
    for s in range(3, 6):
        _grads(d, s)
        od.step()
    _same(a, d, 1e-6)


def test_load_rejects_mismatched_state():
    a = _model()
    oa = TorchStatementAdamW(a.parameters(), lr=0.01)
    _grads(a, 0)
    oa.step()
    sd = oa.state_dict()
    sd["state"][2]["step"] = torch.tensor(7.0)
    with pytest.raises(ValueError, match="different step counts"):
        oa.load_state_dict(sd)
    sd = oa.state_dict()
    sd["param_groups"][0]["params"] = sd["param_groups"][0]["params"][:-1]
    with pytest.raises(ValueError, match="one parameter group"):
        oa.load_state_dict(sd)


def test_step_schedule_accepts_torch_scheduler_state():
    m = _model()
    opt = torch.optim.AdamW(m.parameters(), lr=0.006)
    ref = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[3, 6], gamma=0.1)
    for _ in range(4):
        opt.step()
        ref.step()
    opt2 = torch.optim.AdamW(_model().parameters(), lr=0.006)
    sch = StepSchedule(opt2, "MultiStepLR", total_steps=10, milestones=[0.3, 0.6], gamma=0.1)
    sch.load_state_dict(ref.state_dict())  # carries last_epoch, not last_step
    assert sch.last_step == 4 and sch.get_last_lr() == pytest.approx(ref.get_last_lr())
    assert sch.state_dict()["last_epoch"] == 4
    with pytest.raises(KeyError):
        sch.load_state_dict({"foo": 1})
