"""oracle/pointops_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU front-end of the oracle: the reference's `pointops` Python API
(libs/pointops/functions/__init__.py:1-14) restated on CPU torch tensors on top
of oracle/pointops_oracle.c.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

Parity status: "parity unpinned" vs. a CUDA build (see pointops_oracle.c header);
the pure-torch wrappers (`grouping`, `interpolation`) ARE pinned: tests/golden/
holds outputs of the reference's own Python functions executed in the build
container (tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile the C restatement (gcc; a few hundred ms)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, f))
        for f in ("liboracle_pointops.so", "liboracle_pointops_mt.so")
    )
    if need:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib(mt=False):
    name = "liboracle_pointops_mt.so" if mt else "liboracle_pointops.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _LIBS[name] = ctypes.CDLL(path)
    return _LIBS[name]


def _f(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def _zeros_f(*shape):
    a = np.zeros(shape, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _zeros_i(*shape):
    a = np.zeros(shape, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


# ----------------------------------------------------------------- queries --
def knn_query_raw(nsample, xyz, offset, new_xyz=None, new_offset=None, pad_with_start=False, mt=False):
    """(idx int32 (m,k), dist2 fp32 (m,k)); libs/pointops/functions/query.py:7-24 minus the sqrt."""
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    m = new_xyz.shape[0]
    xa, xp = _f(xyz)
    na, np_ = _f(new_xyz)
    oa, op = _i(offset)
    noa, nop = _i(new_offset)
    ia, ip = _zeros_i(m, nsample)
    da, dp = _zeros_f(m, nsample)
    rc = lib(mt).oracle_knn_query(m, nsample, xp, np_, op, nop, ip, dp, int(pad_with_start))
    assert rc == 0
    return torch.from_numpy(ia), torch.from_numpy(da)


def knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None):
    idx, d2 = knn_query_raw(nsample, xyz, offset, new_xyz, new_offset)
    return idx, torch.sqrt(d2)  # query.py:24


def farthest_point_sampling(xyz, offset, new_offset):
    """libs/pointops/functions/sampling.py:7-27."""
    n, b = xyz.shape[0], offset.shape[0]
    off = [int(v) for v in offset]
    n_max = off[0]
    for i in range(1, b):
        n_max = max(off[i] - off[i - 1], n_max)
    m = int(new_offset[b - 1])
    xa, xp = _f(xyz)
    oa, op = _i(offset)
    noa, nop = _i(new_offset)
    ia, ip = _zeros_i(m)
    tmp = np.full((n,), 1e10, dtype=np.float32)
    rc = lib().oracle_farthest_point_sampling(
        b, n_max, xp, op, nop, tmp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ip
    )
    assert rc == 0
    return torch.from_numpy(ia)


# ----------------------------------------------------- pure-torch wrappers --
def grouping(idx, feat, xyz, new_xyz=None, with_xyz=False):
    """libs/pointops/functions/grouping.py:36-60: -1 selects an appended zero row;
    relative xyz is masked by sign(idx+1)."""
    if new_xyz is None:
        new_xyz = xyz
    m, ns = idx.shape
    c = feat.shape[1]
    lidx = idx.reshape(-1).long()
    valid = (lidx >= 0).to(feat.dtype).unsqueeze(1)
    safe = lidx.clamp(min=0)
    g_feat = (feat[safe] * valid).view(m, ns, c)
    if not with_xyz:
        return g_feat
    g_xyz = (xyz[safe] * valid).view(m, ns, 3) - new_xyz.unsqueeze(1)
    g_xyz = g_xyz * torch.sign(idx + 1).to(xyz.dtype).unsqueeze(-1)
    return torch.cat((g_xyz, g_feat), -1)


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """libs/pointops/functions/interpolation.py:8-22 (idx -1 wraps to the last row, as there)."""
    idx, dist = knn_query(k, xyz, offset, new_xyz, new_offset)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=1, keepdim=True)
    weight = dist_recip / norm
    new_feat = torch.zeros(new_xyz.shape[0], feat.shape[1], dtype=feat.dtype)
    for i in range(k):
        new_feat = new_feat + feat[idx[:, i].long(), :] * weight[:, i].unsqueeze(-1)
    return new_feat


def offset2batch(offset):
    """pointcept/models/utils.py:11-24 without the Python loop."""
    off = offset.long()
    counts = torch.diff(off, prepend=off.new_zeros(1))
    return torch.repeat_interleave(torch.arange(off.numel()), counts)


def batch2offset(batch):
    return torch.cumsum(batch.bincount(), dim=0).long()


# -------------------------------------------------- autograd.Function ops --
class _Grouping2(torch.autograd.Function):
    """libs/pointops/functions/grouping.py:7-33."""

    @staticmethod
    def forward(ctx, input, idx):
        m, ns = idx.shape
        n, c = input.shape
        a, ap = _f(input)
        ia, ip = _i(idx)
        o, op = _zeros_f(m, ns, c)
        lib().oracle_grouping_forward(m, ns, c, ap, ip, op)
        ctx.n = n
        ctx.save_for_backward(idx)
        return torch.from_numpy(o)

    @staticmethod
    def backward(ctx, go):
        (idx,) = ctx.saved_tensors
        m, ns, c = go.shape
        g, gp = _f(go)
        ia, ip = _i(idx)
        o, op = _zeros_f(ctx.n, c)
        lib().oracle_grouping_backward(m, ns, c, gp, ip, op)
        return torch.from_numpy(o), None


grouping2 = _Grouping2.apply


def interpolation_weights(xyz, new_xyz, offset, new_offset, k=3):
    idx, dist = knn_query(k, xyz, offset, new_xyz, new_offset)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=1, keepdim=True)
    return idx, dist_recip / norm


class _Interpolation2(torch.autograd.Function):
    """libs/pointops/functions/interpolation.py:25-56."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, input, offset, new_offset, k=3):
        idx, weight = interpolation_weights(xyz, new_xyz, offset, new_offset, k)
        n, c, m = new_xyz.shape[0], input.shape[1], input.shape[0]
        a, ap = _f(input)
        ia, ip = _i(idx)
        w, wp = _f(weight)
        o, op = _zeros_f(n, c)
        lib().oracle_interpolation_forward(n, c, k, ap, ip, wp, op)
        ctx.m, ctx.k = m, k
        ctx.save_for_backward(idx, weight)
        return torch.from_numpy(o)

    @staticmethod
    def backward(ctx, go):
        idx, weight = ctx.saved_tensors
        n, c = go.shape
        g, gp = _f(go)
        ia, ip = _i(idx)
        w, wp = _f(weight)
        o, op = _zeros_f(ctx.m, c)
        lib().oracle_interpolation_backward(n, c, ctx.k, gp, ip, wp, op)
        return None, None, torch.from_numpy(o), None, None, None


interpolation2 = _Interpolation2.apply


class _Subtraction(torch.autograd.Function):
    """libs/pointops/functions/subtraction.py:7-38."""

    @staticmethod
    def forward(ctx, input1, input2, idx):
        n, c = input1.shape
        ns = idx.shape[-1]
        a, ap = _f(input1)
        b, bp = _f(input2)
        ia, ip = _i(idx)
        o, op = _zeros_f(n, ns, c)
        lib().oracle_subtraction_forward(n, ns, c, ap, bp, ip, op)
        ctx.save_for_backward(idx)
        return torch.from_numpy(o)

    @staticmethod
    def backward(ctx, go):
        (idx,) = ctx.saved_tensors
        n, ns, c = go.shape
        g, gp = _f(go)
        ia, ip = _i(idx)
        o1, o1p = _zeros_f(n, c)
        o2, o2p = _zeros_f(n, c)
        lib().oracle_subtraction_backward(n, ns, c, ip, gp, o1p, o2p)
        return torch.from_numpy(o1), torch.from_numpy(o2), None


subtraction = _Subtraction.apply


class _Aggregation(torch.autograd.Function):
    """libs/pointops/functions/aggregation.py:7-57."""

    @staticmethod
    def forward(ctx, input, position, weight, idx):
        n, ns, c = position.shape
        w_c = weight.shape[-1]
        a, ap = _f(input)
        p, pp = _f(position)
        w, wp = _f(weight)
        ia, ip = _i(idx)
        o, op = _zeros_f(n, c)
        lib().oracle_aggregation_forward(n, ns, c, w_c, ap, pp, wp, ip, op)
        ctx.save_for_backward(input, position, weight, idx)
        return torch.from_numpy(o)

    @staticmethod
    def backward(ctx, go):
        input, position, weight, idx = ctx.saved_tensors
        n, ns, c = position.shape
        w_c = weight.shape[-1]
        a, ap = _f(input)
        p, pp = _f(position)
        w, wp = _f(weight)
        ia, ip = _i(idx)
        g, gp = _f(go)
        gi, gip = _zeros_f(n, c)
        gpos, gposp = _zeros_f(n, ns, c)
        gw, gwp = _zeros_f(n, ns, w_c)
        lib().oracle_aggregation_backward(n, ns, c, w_c, ap, pp, wp, ip, gp, gip, gposp, gwp)
        return torch.from_numpy(gi), torch.from_numpy(gpos), torch.from_numpy(gw), None


aggregation = _Aggregation.apply


class _AttentionRelationStep(torch.autograd.Function):
    """libs/pointops/functions/attention.py:12-63 (grad_weight is computed by the
    kernel but the wrapper returns None for it, :63)."""

    @staticmethod
    def forward(ctx, query, key, weight, index_target, index_refer):
        _, g, c = query.shape
        m = index_target.shape[0]
        q, qp = _f(query)
        k, kp = _f(key)
        w, wp = _f(weight)
        t, tp = _i(index_target)
        r, rp = _i(index_refer)
        o, op = _zeros_f(m, g)
        lib().oracle_attention_relation_step_forward(m, g, c, qp, kp, wp, tp, rp, op)
        ctx.save_for_backward(query, key, weight, index_target, index_refer)
        return torch.from_numpy(o)

    @staticmethod
    def backward(ctx, go):
        query, key, weight, index_target, index_refer = ctx.saved_tensors
        n, g, c = query.shape
        m = index_target.shape[0]
        q, qp = _f(query)
        k, kp = _f(key)
        w, wp = _f(weight)
        t, tp = _i(index_target)
        r, rp = _i(index_refer)
        gg, ggp = _f(go)
        gq, gqp = _zeros_f(n, g, c)
        gk, gkp = _zeros_f(n, g, c)
        gw, gwp = _zeros_f(c)
        lib().oracle_attention_relation_step_backward(m, g, c, qp, gqp, kp, gkp, wp, gwp, tp, rp, ggp)
        return torch.from_numpy(gq), torch.from_numpy(gk), None, None, None


attention_relation_step = _AttentionRelationStep.apply


class _AttentionFusionStep(torch.autograd.Function):
    """libs/pointops/functions/attention.py:66-117."""

    @staticmethod
    def forward(ctx, weight, value, index_target, index_refer):
        n, g, c = value.shape
        m = index_refer.shape[0]
        w, wp = _f(weight)
        v, vp = _f(value)
        t, tp = _i(index_target)
        r, rp = _i(index_refer)
        o, op = _zeros_f(n, g, c)
        lib().oracle_attention_fusion_step_forward(m, g, c, wp, vp, tp, rp, op)
        ctx.save_for_backward(weight, value, index_target, index_refer)
        return torch.from_numpy(o)

    @staticmethod
    def backward(ctx, go):
        weight, value, index_target, index_refer = ctx.saved_tensors
        n, g, c = value.shape
        m = index_target.shape[0]
        w, wp = _f(weight)
        v, vp = _f(value)
        t, tp = _i(index_target)
        r, rp = _i(index_refer)
        gg, ggp = _f(go)
        gw, gwp = _zeros_f(m, g)
        gv, gvp = _zeros_f(n, g, c)
        lib().oracle_attention_fusion_step_backward(m, g, c, wp, gwp, vp, gvp, tp, rp, ggp)
        return torch.from_numpy(gw), torch.from_numpy(gv), None, None


attention_fusion_step = _AttentionFusionStep.apply


def knn_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None,
                        nsample=None, with_xyz=False):
    """libs/pointops/functions/utils.py:5-19."""
    if idx is None:
        idx, _ = knn_query(nsample, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx
