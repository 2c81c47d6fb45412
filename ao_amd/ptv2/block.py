"""A whole PT-v2m2 Block (point_transformer_v2m2_base.py:131-177) as one native call per direction.

`Block.forward` of model.py dispatches here when the block is in the shape the native runtime covers
(fp32 CUDA rows, pe_bias only, no attention dropout in training, channels/groups the GVA kernels are
instantiated for).  The module keeps its nn.Parameters / buffers under the reference's state_dict
names; this file only gathers their device pointers in the order of `PTV2_BLK_*` (include/ptv2_hip.h)
and hands them to ao_amd/csrc/block.hip.  Parameter gradients come back in one flat buffer that is
split into per-parameter views for autograd (DDP / the optimizer see ordinary `.grad`s).
"""
import ctypes

import torch

from .. import _lib
from . import gva as _gva

NPARAM, NBN = 30, 7
_P = ctypes.c_void_p

_lib.register({
    "rows_gemm_hip_launcher": (_lib._c_int, [_lib._c_int] * 3 + [_lib._vp, _lib._vp, _lib._c_int, _lib._vp, _lib._vp,
                                                                 _lib._c_int, _lib._vp]),
    "rows_gemm_multi_hip_launcher": (_lib._c_int, [_lib._c_int] * 5 + [_lib._vp, _lib._vp, _lib._c_int, _lib._vp, _lib._vp,
                                                                       _lib._c_int, _lib._vp]),
    "rows_gemm_fused_hip_launcher": (_lib._c_int, [_lib._c_int] * 5 + [_lib._vp, _lib._vp, _lib._c_int, _lib._vp, _lib._vp,
                                                                       _lib._c_int, _lib._vp, _lib._vp, _lib._vp, _lib._vp]),
    "ptv2_block_saved_bytes": (_lib._c_size, [_lib._c_int] * 4),
    "ptv2_block_workspace_bytes": (_lib._c_size, [_lib._c_int] * 4),
    "ptv2_block_param_layout": (_lib._c_int, [_lib._c_int, _lib._c_int, _lib._vp]),
    "ptv2_block_forward_hip_launcher": (_lib._c_int, [_lib._vp, _lib._vp, _lib._c_size, _lib._vp]),
    "ptv2_block_backward_hip_launcher": (_lib._c_int, [_lib._vp, _lib._vp, _lib._vp, _lib._c_size, _lib._vp]),
})


class _Blk(ctypes.Structure):  # mirrors ptv2_block
    _fields_ = ([(n_, ctypes.c_int) for n_ in ("n", "k", "c", "g", "training")]
                + [("eps", ctypes.c_float), ("momentum", ctypes.c_float)]
                + [(n_, _P) for n_ in ("x", "coord", "idx", "mu", "cov", "rowscale")]
                + [("param", _P * NPARAM), ("run_mean", _P * NBN), ("run_var", _P * NBN), ("batches", _P * NBN),
                   ("y", _P), ("saved", _P), ("saved_bytes", ctypes.c_size_t), ("matmul_bf16", ctypes.c_int),
                   ("attn_drop_p", ctypes.c_float), ("attn_drop_seed", ctypes.c_uint)])


class _BlkGrads(ctypes.Structure):  # mirrors ptv2_block_grads
    _fields_ = [(n_, _P) for n_ in ("gy", "inv_ptr", "inv_rows", "gx", "gparam")] + [("gp", _P * NPARAM)]


_lib.check_struct(0, _Blk)
_lib.check_struct(1, _BlkGrads)


def rows_gemm(x, w, bias=None, w_kmajor=False, out=None, accumulate=False):
    """Y = X op(W) + bias on the fp32 MFMA kernel (csrc/gemm.hip).  w_kmajor=False: W (n,k), y = x W^T;
    w_kmajor=True: W (k,n), y = x W."""
    _lib.require_cuda(x, w)
    x, w = x.contiguous(), w.contiguous()
    m, k = x.shape
    n = w.shape[1] if w_kmajor else w.shape[0]
    assert (w.shape[0] if w_kmajor else w.shape[1]) == k
    if out is None:
        assert not accumulate
        out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    rc = _lib.lib().rows_gemm_hip_launcher(m, n, k, x.data_ptr(), w.data_ptr(), int(w_kmajor), _lib.ptr(bias),
                                           out.data_ptr(), int(accumulate), _lib.stream_ptr())
    _lib.check(rc, "rows_gemm_hip_launcher")
    return out


def block_params(blk):
    """The Block's parameters in PTV2_BLK_* order (None for absent biases) and its 7 BatchNorms."""
    a = blk.attn
    lq, lk, lv = a.linear_q, a.linear_k, a.linear_v
    pb, we = a.linear_p_bias, a.weight_encoding
    bns = [blk.norm1.norm, lq[1].norm, lk[1].norm, pb[1].norm, we[1].norm, blk.norm2.norm, blk.norm3.norm]
    params = [blk.fc1.weight, bns[0].weight, bns[0].bias,
              lq[0].weight, lq[0].bias, bns[1].weight, bns[1].bias,
              lk[0].weight, lk[0].bias, bns[2].weight, bns[2].bias,
              lv.weight, lv.bias,
              pb[0].weight, pb[0].bias, bns[3].weight, bns[3].bias, pb[3].weight, pb[3].bias,
              we[0].weight, we[0].bias, bns[4].weight, bns[4].bias, we[3].weight, we[3].bias,
              bns[5].weight, bns[5].bias, blk.fc3.weight, bns[6].weight, bns[6].bias]
    assert len(params) == NPARAM
    return params, bns


_layout_cache = {}


def param_layout(c, g):
    key = (c, g)
    if key not in _layout_cache:
        off = (ctypes.c_longlong * (NPARAM + 1))()
        _lib.check(_lib.lib().ptv2_block_param_layout(c, g, ctypes.addressof(off)), "ptv2_block_param_layout")
        _layout_cache[key] = list(off)
    return _layout_cache[key]


def _slots(blk):
    """(owner dict, name) of every tensor whose device pointer goes into the ptv2_block struct, in ABI order: the 30
    parameter slots, then (running_mean, running_var, num_batches_tracked) of the 7 BatchNorms.  The dicts are the
    modules' own `_parameters` / `_buffers`, so looking a slot up again sees re-assigned Parameters and buffers
    (load_state_dict(assign=True), `bn.running_mean = ...`) without walking the module tree."""
    params, bns = block_params(blk)
    owners = {}
    for m in blk.modules():
        for name, t in m._parameters.items():
            if t is not None:
                owners[id(t)] = (m._parameters, name)
    slots = [owners[id(p)] if p is not None else None for p in params]
    for bn in bns:
        for name in ("running_mean", "running_var", "num_batches_tracked"):
            slots.append((bn._buffers, name) if bn._buffers.get(name) is not None else None)
    return slots


def _pointer_key(slots):
    return tuple(0 if s is None else s[0][s[1]].data_ptr() for s in slots)


class _Plan:
    """Per-Block cache of everything that does not change between steps: the parameter / BatchNorm objects in
    ABI order, the flat-gradient layout and a ptv2_block struct with the parameter pointers filled in.  The key is
    the device pointer of EVERY parameter and BatchNorm buffer (51 of them, looked up through the owning modules'
    dicts: ~8 us per call): the plan is rebuilt when any of them moves -- an optimizer that re-homes `p.data`
    (FlatAdamW), `.to()`, `load_state_dict(assign=True)`, a replaced buffer."""

    def __init__(self, blk):
        a = blk.attn
        self.params, self.bns = block_params(blk)
        self.c, self.g = a.embed_channels, a.groups
        b0 = self.bns[0]
        # static_core: what the whole-model runtime needs of a Block (it implements enable_checkpoint itself,
        # ptv2_model.checkpoint); static_ok: the per-Block launchers, which leave checkpointing to torch
        self.static_core = (
            not a.pe_multiplier and a.pe_bias
            and all(p is None or (p.dtype == torch.float32 and p.is_cuda and p.is_contiguous()) for p in self.params)
            and all(bn.momentum is not None and bn.affine and bn.eps == b0.eps and bn.momentum == b0.momentum
                    and (bn.running_mean is None) == (b0.running_mean is None) for bn in self.bns))
        self.static_ok = self.static_core and not blk.enable_checkpoint
        self.has_running = b0.running_mean is not None
        self.off = param_layout(self.c, self.g)
        self.slots = [self.off[i + 1] - self.off[i] for i in range(NPARAM)]
        self.tensor_slots = _slots(blk)
        self.key = _pointer_key(self.tensor_slots)
        args = _Blk()
        args.c, args.g = self.c, self.g
        args.eps, args.momentum = float(b0.eps), float(b0.momentum if b0.momentum is not None else 0.1)
        for i, p in enumerate(self.params):
            args.param[i] = p.data_ptr() if p is not None else None
        for i, bn in enumerate(self.bns):
            has = bn.running_mean is not None
            args.run_mean[i] = bn.running_mean.data_ptr() if has else None
            args.run_var[i] = bn.running_var.data_ptr() if has else None
            args.batches[i] = bn.num_batches_tracked.data_ptr() if has and bn.track_running_stats else None
        self.args = args


def plan(blk):
    p = blk.__dict__.get("_ao_plan")
    if p is not None:
        try:
            fresh = p.key == _pointer_key(p.tensor_slots)
        except (KeyError, AttributeError):  # a slot disappeared (parameter set to None): rebuild
            fresh = False
        if fresh:
            return p
    p = _Plan(blk)
    blk.__dict__["_ao_plan"] = p
    return p


def supported(blk, feat, idx):
    if not (feat.is_cuda and feat.dtype in (torch.float32, torch.bfloat16, torch.float16) and feat.dim() == 2
            and feat.shape[0] >= 2):
        return False
    p = plan(blk)
    if not p.static_ok or (not blk.training and not p.has_running):
        return False
    if blk.attn.attn_drop_rate != 0.0 and blk.training and not _gva.dropout_supported(p.c, p.g, idx.shape[1]):
        return False
    return _gva.supported(p.c, p.g, idx.shape[1])


def _fill(p, x, coord, idx, mu, cov, rowscale, y, saved, training, bf16=False, drop=(0.0, 0)):
    args = p.args
    args.matmul_bf16 = int(bf16)
    args.attn_drop_p, args.attn_drop_seed = float(drop[0]), int(drop[1])
    args.n, args.k = idx.shape
    args.training = int(training)
    args.x, args.coord, args.idx = x.data_ptr(), coord.data_ptr(), idx.data_ptr()
    args.mu, args.cov, args.rowscale = _lib.ptr(mu), _lib.ptr(cov), _lib.ptr(rowscale)
    args.y, args.saved, args.saved_bytes = y.data_ptr(), saved.data_ptr(), saved.numel()
    return args


class _NativeBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, coord, idx, mu, cov, inv, rowscale, training, bf16, drop, *params):
        x = x.contiguous()
        n, k = idx.shape
        dev = x.device
        L = _lib.lib()
        y = torch.empty((n, p.c), dtype=torch.float32, device=dev)
        saved = torch.empty(L.ptv2_block_saved_bytes(n, k, p.c, p.g), dtype=torch.uint8, device=dev)
        args = _fill(p, x, coord, idx, mu, cov, rowscale, y, saved, training, bf16, drop)
        ws = _lib.workspace(L.ptv2_block_workspace_bytes(n, k, p.c, p.g), dev)
        rc = L.ptv2_block_forward_hip_launcher(ctypes.addressof(args), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "ptv2_block_forward_hip_launcher")
        ctx.save_for_backward(x, coord, idx, mu, cov, rowscale, y, saved)
        ctx.plan, ctx.inv, ctx.training, ctx.bf16, ctx.drop = p, inv, training, bf16, drop
        return y

    @staticmethod
    def backward(ctx, gy):
        x, coord, idx, mu, cov, rowscale, y, saved = ctx.saved_tensors
        p = ctx.plan
        n, k = idx.shape
        dev = x.device
        L = _lib.lib()
        gy = gy.contiguous()
        args = _fill(p, x, coord, idx, mu, cov, rowscale, y, saved, ctx.training, ctx.bf16, ctx.drop)  # (the forward's mask again)
        gx = torch.empty_like(x)
        gflat = torch.empty(p.off[NPARAM], dtype=torch.float32, device=dev)
        inv_ptr, inv_rows = ctx.inv if ctx.inv is not None else _gva.inverse_table(idx)
        grads = _BlkGrads()
        grads.gy, grads.gx, grads.gparam = gy.data_ptr(), gx.data_ptr(), gflat.data_ptr()
        grads.inv_ptr, grads.inv_rows = _lib.ptr(inv_ptr), _lib.ptr(inv_rows)
        ws = _lib.workspace(L.ptv2_block_workspace_bytes(n, k, p.c, p.g), dev)
        rc = L.ptv2_block_backward_hip_launcher(ctypes.addressof(args), ctypes.addressof(grads), ws.data_ptr(), ws.numel(),
                                                _lib.stream_ptr())
        _lib.check(rc, "ptv2_block_backward_hip_launcher")
        gp = []
        for chunk, prm, slot in zip(gflat.split_with_sizes(p.slots), p.params, p.slots):
            if prm is None:
                gp.append(None)
                continue
            if slot != prm.numel():
                chunk = chunk[:prm.numel()]
            gp.append(chunk if prm.dim() == 1 else chunk.view(prm.shape))
        return (gx, None, None, None, None, None, None, None, None, None, None, *gp)


def block_forward(blk, feat, coord, idx, rowscale):
    """Block.forward on the native runtime (call `supported` first)."""
    p = plan(blk)
    training = blk.training or not p.has_running
    mu = cov = None
    if training:
        mu, cov = _gva._pos_moments(_gva._HipImpl, coord, idx)
    inv = _gva.inverse_table(idx) if torch.is_grad_enabled() else None
    # under torch.autocast the Linear products run on the bf16 matrix cores (native_model.matmul_bf16); BatchNorm,
    # softmax, accumulation and the activations in memory stay fp32, as autocast keeps them
    from .native_model import matmul_bf16

    bf16 = matmul_bf16()
    rate = blk.attn.attn_drop_rate
    drop = (rate, _gva.next_drop_seed()) if (blk.training and rate > 0.0) else (0.0, 0)
    with torch.autocast("cuda", enabled=False):
        return _NativeBlock.apply(feat.float(), p, coord, idx, mu, cov, inv, rowscale, training, bf16, drop, *p.params)
