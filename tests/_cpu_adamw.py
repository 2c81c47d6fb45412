"""Test helper: FlatAdamW with the HIP update kernel replaced by a torch statement of torch.optim.AdamW's arithmetic,
so that the HOST logic of the class (flat views, gradient flattening, grad_scale, state_dict layout) can be driven on
CPU tensors.  Lives under tests/ -- the package itself has no CPU arithmetic (FlatAdamW._launch raises off-GPU)."""
import math

import torch

from ao_amd.ptv2.optim import FlatAdamW


class TorchStatementAdamW(FlatAdamW):
    def _launch(self, g, grp, grad_scale):
        b1, b2 = grp["betas"]
        lr, eps, wd, t = grp["lr"], grp["eps"], grp["weight_decay"], self._step
        g = g * grad_scale
        p, m, v = self.flat_param, self.exp_avg, self.exp_avg_sq
        p.mul_(1 - lr * wd)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))
