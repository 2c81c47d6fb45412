"""AdamW over one flat parameter buffer (ao_amd/csrc/optim.hip).

`FlatAdamW(params, lr, betas, eps, weight_decay)` moves the parameters into one contiguous fp32 buffer (each
`param.data` becomes a view of it, so modules, state_dict and checkpoints are unaffected) and updates all of them
with one streaming kernel per step instead of torch's 24 multi-tensor launches over 840 tensors.  It is a
`torch.optim.Optimizer` (one param group; `param_groups[0]["lr"]` is read every step, so LR schedulers work);
the arithmetic is torch.optim.AdamW's.  Build it AFTER the model is on its device; `.to()` afterwards would
detach the parameters from the buffer.

Checkpoints: `state_dict()` / `load_state_dict()` speak torch.optim.AdamW's layout (state[i] = {step, exp_avg,
exp_avg_sq} per parameter in group order, pointcept/engines/hooks/misc.py:180-183 saves it, :247 restores it), so a
run resumes with its moments and bias-correction step, and an optimizer state written by the reference's
torch.optim.AdamW loads into this class (and the other way round).
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
        assert all(p.device == dev and p.dtype == torch.float32 for p in self._params)
        # every parameter owns a slot aligned to 4 floats (the gradient kernels store float4): the layout of
        # native_model.grad_layout, so that the native model backward can write its gradients straight into a buffer
        # this optimizer consumes without a copy
        self._sizes = [p.numel() for p in self._params]
        self._offsets, off = [], 0
        for k in self._sizes:
            self._offsets.append(off)
            off += (k + 3) // 4 * 4
        self._n = off
        flat = torch.zeros(self._n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, k, o in zip(self._params, self._sizes, self._offsets):
                view = flat[o:o + k].view_as(p)
                view.copy_(p.data)
                p.data = view
        self.flat_param = flat
        self.flat_grad = torch.zeros_like(flat)
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self._step = 0

    def _slots(self, flat):
        """Per-parameter 1-D views of a flat buffer in this optimizer's layout."""
        return [flat[o:o + k] for k, o in zip(self._sizes, self._offsets)]

    def flatten_grads(self):
        """The gradients as one flat buffer in this optimizer's layout.  When every `.grad` already aliases one such
        buffer (the native model backward writes them that way, ao_amd/ptv2/native_model.py) that buffer is returned
        as it is; otherwise one multi-tensor copy gathers them (missing gradients count as zero)."""
        g0 = self._params[0].grad
        if g0 is not None and g0.is_contiguous():
            base = g0.data_ptr() - 4 * self._offsets[0]
            try:
                aliased = all(p.grad.data_ptr() == base + 4 * o for p, o in zip(self._params, self._offsets))
            except AttributeError:  # a parameter without a gradient
                aliased = False
            start = g0.storage_offset() - self._offsets[0]
            if aliased and start >= 0 and g0.untyped_storage().nbytes() >= 4 * (start + self._n):
                return torch.empty(0, dtype=torch.float32, device=g0.device).set_(g0.untyped_storage(), start, (self._n,), (1,))
        views = self._slots(self.flat_grad)
        dst = [v for v, p in zip(views, self._params) if p.grad is not None]
        src = [p.grad.reshape(-1) for p in self._params if p.grad is not None]
        if len(dst) != len(self._params):
            self.flat_grad.zero_()
        torch._foreach_copy_(dst, src)
        return self.flat_grad

    # -- checkpoint format: torch.optim.AdamW's --
    def state_dict(self):
        """{"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]} exactly as torch.optim.AdamW lays
        it out (tensors are copies shaped like their parameter; "step" is a float scalar tensor as in torch >= 1.12)."""
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self._params)))
        state = {}
        if self._step > 0:
            m, v = self._slots(self.exp_avg), self._slots(self.exp_avg_sq)
            for i, p in enumerate(self._params):
                state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": m[i].view_as(p).clone(),
                            "exp_avg_sq": v[i].view_as(p).clone()}
        return {"state": state, "param_groups": [group]}

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self._params):
            raise ValueError("FlatAdamW.load_state_dict: expected one parameter group of %d parameters, got %s"
                             % (len(self._params), [len(g["params"]) for g in groups]))
        for k, v in groups[0].items():
            if k != "params" and k in self.param_groups[0]:  # torch-only keys (amsgrad, foreach, fused, ...) are dropped
                self.param_groups[0][k] = v
        if groups[0].get("amsgrad") or groups[0].get("maximize"):
            raise ValueError("FlatAdamW does not implement amsgrad / maximize")
        ids = groups[0]["params"]
        state = state_dict["state"]
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        m, v = self._slots(self.exp_avg), self._slots(self.exp_avg_sq)
        for i, (pid, p) in enumerate(zip(ids, self._params)):
            st = state.get(pid, state.get(str(pid)))
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("FlatAdamW.load_state_dict: parameter %d has shape %s, state has %s"
                                 % (i, tuple(p.shape), tuple(st["exp_avg"].shape)))
            m[i].copy_(st["exp_avg"].reshape(-1))
            v[i].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FlatAdamW.load_state_dict: parameters carry different step counts %s (one kernel applies "
                             "one bias correction)" % sorted(steps))
        if steps and len(state) != len(self._params):
            raise ValueError("FlatAdamW.load_state_dict: state for %d of %d parameters" % (len(state), len(self._params)))
        self._step = steps.pop() if steps else 0

    def _launch(self, g, grp, grad_scale):
        """The update itself: ao_amd/csrc/optim.hip on the flat buffers.  GPU only -- there is no CPU arithmetic in this
        package (tests/test_ddp_cpu.py substitutes a torch statement of the same update to drive the host logic)."""
        if not self.flat_param.is_cuda:
            raise RuntimeError("FlatAdamW updates on the GPU only (parameters are on %s); there is no CPU fallback"
                               % self.flat_param.device)
        rc = _lib.lib().adamw_flat_hip_launcher(self._n, self.flat_param.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                                                self.exp_avg_sq.data_ptr(), float(grp["lr"]), float(grp["betas"][0]),
                                                float(grp["betas"][1]), float(grp["eps"]), float(grp["weight_decay"]),
                                                self._step, float(grad_scale), _lib.stream_ptr())
        _lib.check(rc, "adamw_flat_hip_launcher")

    @torch.no_grad()
    def step(self, closure=None, flat_grad=None, grad_scale=1.0):
        """flat_grad: an already flattened (e.g. all-reduced) gradient in this optimizer's parameter order; default:
        gather the `.grad`s.  grad_scale multiplies it inside the kernel (1 / world for a summed gradient)."""
        loss = closure() if closure is not None else None
        g = self.flatten_grads() if flat_grad is None else flat_grad
        self._step += 1
        self._launch(g, self.param_groups[0], grad_scale)
        return loss
