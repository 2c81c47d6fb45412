"""Optimizer / learning-rate schedule construction with the reference's config vocabulary
(pointcept/utils/optimizer.py:20-55, pointcept/utils/scheduler.py:14-145; used by engines/train.py:275-283).

`build_optimizer(cfg, model)` returns FlatAdamW (one HIP kernel per step, ao_amd/ptv2/optim.py) for
`dict(type="AdamW", ...)` without per-keyword parameter groups, torch's optimizer otherwise.
`build_scheduler(cfg, optimizer)` returns a StepSchedule: the closed form of the reference's scheduler of that name
(milestones / warm-up / cycle lengths given as FRACTIONS of `total_steps`), stepped once after every optimizer step.
It writes `lr` (and, for OneCycleLR, beta1 / momentum) into every param group, which FlatAdamW reads each step.
"""
import math

import torch

from .optim import FlatAdamW


def build_optimizer(cfg, model, param_dicts=None, flat=True):
    cfg = dict(cfg)
    kind = cfg.pop("type")
    if param_dicts:
        # optimizer.py:23-46: group 0 = everything no keyword claims; group i+1 = parameters whose name contains
        # param_dicts[i].keyword, with that entry's lr / momentum / weight_decay
        groups = [dict(params=[], lr=cfg["lr"])] + [
            dict(params=[], **{k: d[k] for k in ("lr", "momentum", "weight_decay") if k in d}) for d in param_dicts]
        for name, p in model.named_parameters():
            hit = next((i for i, d in enumerate(param_dicts) if d["keyword"] in name), -1)
            groups[hit + 1]["params"].append(p)
        params = [g for g in groups if g["params"]]
    else:
        params = list(model.parameters())
    if kind == "AdamW" and flat and not param_dicts and all(p.is_cuda for p in params):
        return FlatAdamW(params, **cfg)
    return {"SGD": torch.optim.SGD, "Adam": torch.optim.Adam, "AdamW": torch.optim.AdamW}[kind](params, **cfg)


def _anneal_cos(a, b, p):
    return b + (a - b) / 2.0 * (math.cos(math.pi * p) + 1)


def _anneal_linear(a, b, p):
    return (b - a) * p + a


class StepSchedule:
    """factor(s) * base_lr for optimizer step s = 0, 1, ...; `step()` after every optimizer step."""

    KINDS = ("MultiStepLR", "MultiStepWithWarmupLR", "PolyLR", "ExpLR", "CosineAnnealingLR", "OneCycleLR")

    def __init__(self, optimizer, type, total_steps, **kw):
        if type not in self.KINDS:
            raise KeyError("unknown scheduler type %r" % (type,))
        self.optimizer, self.kind, self.total_steps, self.kw = optimizer, type, int(total_steps), kw
        self.last_step = 0
        if type == "OneCycleLR":
            max_lr = kw["max_lr"]
            n = len(optimizer.param_groups)
            self.max_lrs = list(max_lr) if isinstance(max_lr, (list, tuple)) else [max_lr] * n
            self.cycle_momentum = kw.get("cycle_momentum", True)
        else:
            self.base_lrs = [g.setdefault("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self._apply()

    # -- closed forms (oracle/host_ref.py:lr_curve restates the same arithmetic for the tests) --
    def _multiplier(self, s):
        kw, total = self.kw, self.total_steps
        if self.kind == "MultiStepLR":
            # torch's MultiStepLR looks the integer step up among the (float) milestones: a milestone that is not a
            # whole number of steps never fires -- kept, it is what the reference trains with
            miles = [r * total for r in kw["milestones"]]
            return kw.get("gamma", 0.1) ** sum(1 for t in range(1, s + 1) if float(t) in miles)
        if self.kind == "MultiStepWithWarmupLR":
            f = 1.0
            for r in kw["milestones"]:
                if s < r * total:
                    break
                f *= kw.get("gamma", 0.1)
            wr, ws = kw.get("warmup_rate", 0.05), kw.get("warmup_scale", 1e-6)
            return (1 - (1 - s / wr / total) * (1 - ws) if s <= wr * total else 1.0) * f
        if self.kind == "PolyLR":
            return (1 - s / (total + 1)) ** kw.get("power", 0.9)
        if self.kind == "ExpLR":
            return kw.get("gamma", 0.9) ** (s / total)
        raise AssertionError

    def values(self, s):
        """(lr per group, momentum per group or None) in effect for optimizer step s."""
        kw, total = self.kw, self.total_steps
        if self.kind == "OneCycleLR":
            if s > total:
                raise ValueError("OneCycleLR stepped %d times, total_steps is %d" % (s, total))
            anneal = _anneal_cos if kw.get("anneal_strategy", "cos") == "cos" else _anneal_linear
            e1, e2 = float(kw.get("pct_start", 0.3) * total) - 1, total - 1
            m_hi, m_lo = kw.get("max_momentum", 0.95), kw.get("base_momentum", 0.85)
            lrs, moms = [], []
            for mx in self.max_lrs:
                init = mx / kw.get("div_factor", 25.0)
                low = init / kw.get("final_div_factor", 1e4)
                if s <= e1:
                    p = s / e1
                    lrs.append(anneal(init, mx, p)); moms.append(anneal(m_hi, m_lo, p))
                else:
                    p = (s - e1) / (e2 - e1)
                    lrs.append(anneal(mx, low, p)); moms.append(anneal(m_lo, m_hi, p))
            return lrs, (moms if self.cycle_momentum else None)
        if self.kind == "CosineAnnealingLR":
            eta = kw.get("eta_min", 0.0)
            return [eta + (b - eta) * (1 + math.cos(math.pi * s / total)) / 2 for b in self.base_lrs], None
        m = self._multiplier(s)
        return [b * m for b in self.base_lrs], None

    def _apply(self):
        lrs, moms = self.values(self.last_step)
        for i, g in enumerate(self.optimizer.param_groups):
            g["lr"] = lrs[i]
            if moms is not None:
                if "betas" in g:
                    g["betas"] = (moms[i], g["betas"][1])
                elif "momentum" in g:
                    g["momentum"] = moms[i]

    def step(self):
        self.last_step += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        # "last_epoch" is what torch's schedulers call the same counter (number of step() calls so far)
        return {"last_step": self.last_step, "last_epoch": self.last_step}

    def load_state_dict(self, state):
        """Accepts this class's own state or a torch.optim.lr_scheduler state_dict (the reference checkpoints the
        latter, pointcept/engines/hooks/misc.py:184,248): there the counter is `last_epoch`."""
        if "last_step" in state:
            self.last_step = int(state["last_step"])
        elif "last_epoch" in state:
            self.last_step = int(state["last_epoch"])
        else:
            raise KeyError("scheduler state has neither 'last_step' nor 'last_epoch': %s" % sorted(state))
        self._apply()


def build_scheduler(cfg, optimizer, total_steps=None):
    """cfg: the config's `scheduler` dict; total_steps as engines/train.py:281 sets it (len(loader) * eval_epoch)."""
    cfg = dict(cfg)
    if total_steps is not None:
        cfg["total_steps"] = total_steps
    return StepSchedule(optimizer, **cfg)
