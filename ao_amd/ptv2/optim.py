"""AdamW over one flat parameter buffer (ao_amd/csrc/optim.hip).

`FlatAdamW(params, lr, betas, eps, weight_decay)` moves the parameters into one contiguous fp32 buffer (each
`param.data` becomes a view of it, so modules, state_dict and checkpoints are unaffected) and updates all of them
with one streaming kernel per step instead of torch's 24 multi-tensor launches over 840 tensors.  It is a
`torch.optim.Optimizer` (one param group; `param_groups[0]["lr"]` is read every step, so LR schedulers work);
the arithmetic is torch.optim.AdamW's.  Build it AFTER the model is on its device; `.to()` afterwards would
detach the parameters from the buffer.
"""
import torch

from .. import _lib


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        assert len(self.param_groups) == 1, "FlatAdamW takes one parameter group"
        self._params = self.param_groups[0]["params"]
        dev = self._params[0].device
        assert all(p.is_cuda and p.device == dev and p.dtype == torch.float32 for p in self._params)
        self._sizes = [p.numel() for p in self._params]
        total = sum(self._sizes)
        self._n = (total + 3) // 4 * 4
        flat = torch.zeros(self._n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p, k in zip(self._params, self._sizes):
                view = flat[off:off + k].view_as(p)
                view.copy_(p.data)
                p.data = view
                off += k
        self.flat_param = flat
        self.flat_grad = torch.zeros_like(flat)
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self._total = total
        self._step = 0

    def flatten_grads(self):
        """One multi-tensor copy of every `.grad` into the flat gradient buffer (missing gradients count as zero)."""
        views = self.flat_grad[:self._total].split(self._sizes)
        dst = [v for v, p in zip(views, self._params) if p.grad is not None]
        src = [p.grad.reshape(-1) for p in self._params if p.grad is not None]
        if len(dst) != len(self._params):
            self.flat_grad.zero_()
        torch._foreach_copy_(dst, src)
        return self.flat_grad

    @torch.no_grad()
    def step(self, closure=None, flat_grad=None, grad_scale=1.0):
        """flat_grad: an already flattened (e.g. all-reduced) gradient in this optimizer's parameter order; default:
        gather the `.grad`s.  grad_scale multiplies it inside the kernel (1 / world for a summed gradient)."""
        loss = closure() if closure is not None else None
        g = self.flatten_grads() if flat_grad is None else flat_grad
        grp = self.param_groups[0]
        self._step += 1
        rc = _lib.lib().adamw_flat_hip_launcher(self._n, self.flat_param.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                                                self.exp_avg_sq.data_ptr(), float(grp["lr"]), float(grp["betas"][0]),
                                                float(grp["betas"][1]), float(grp["eps"]), float(grp["weight_decay"]),
                                                self._step, float(grad_scale), _lib.stream_ptr())
        _lib.check(rc, "adamw_flat_hip_launcher")
        return loss
