"""Fused grouped vector attention (GVA) for PT-v2m2 on MI355X.

The reference evaluates GroupedVectorAttention.forward (point_transformer_v2m2_base.py:103-129)
as ~25 eager ops that each stream an (N,K,C) fp32 tensor, including a dense (N*K,C)x(C,C) GEMM for
the positional-encoding bias.  Here the same function is re-associated so that nothing of size
N*K*C is ever materialised and the only per-neighbour tensors are of size N*K*G (G = C/8):

  pos[n,s]   = mask * (coord[idx[n,s]] - coord[n])
  P[n,s,:]   = ReLU(BN_p(pos Wp1^T + bp1))        = ReLU(pos a^T + b)      (BN_p folded, see below)
  peb[n,s,:] = P Wp2^T + bp2                                               (never formed)
  W1[n,s,:]  = (k[idx]*mask - q[n] + peb) Ww1^T + bw1
             = kW[idx]*mask - qW[n] + P M + cW,   kW = k Ww1^T, qW = q Ww1^T (N,G),
                                                  M = (Ww1 Wp2)^T (C,G), cW = Ww1 bp2 + bw1
  w          = mask * softmax_s( ReLU(BN_w(W1)) Ww2^T + bw2 )
  out[n,c]   = sum_s w[n,s,g(c)] (v[idx[n,s],c]*mask + peb[n,s,c])
             = out_v[n,c] + sum_c' A[n,g(c),c'] Wp2[c,c'] + bp2[c] sw[n,g(c)],
               A[n,g,:] = sum_s w[n,s,g] P[n,s,:],  sw[n,g] = sum_s w[n,s,g]

BN_p (training) needs the batch statistics of pos Wp1^T + bp1 over all N*K rows; they follow in closed
form from the mean (3) and covariance (3x3) of pos, which depend on the neighbour table only and are
computed once per table by a HIP reduction.  BN_w statistics come out of the logits kernel as per-
channel sums.  All dense products (kW, qW, M, the grouped A x Wp2 product) stay in torch on rocBLAS;
autograd composes the two HIP stages (`logits`, `aggregate`) with them, which also yields the exact
BatchNorm backward through the batch statistics.
"""
import ctypes
import os

import torch

from .. import _lib
from .layers import skinny_linear

_SIG = {
    "gva_pos_stats_hip_launcher": (_lib._c_int, [_lib._c_int] * 2 + [_lib._vp] * 5 + [_lib._c_size, _lib._vp]),
    "gva_pos_moments_hip_launcher": (_lib._c_int, [_lib._c_int] * 2 + [_lib._vp] * 5 + [_lib._c_size, _lib._vp]),
    "gva_logits_forward_hip_launcher": (_lib._c_int, [_lib._c_int] * 4 + [_lib._vp] * 12 + [_lib._c_size, _lib._vp]),
    "gva_logits_backward_hip_launcher": (_lib._c_int, [_lib._c_int] * 4 + [_lib._vp] * 18 + [_lib._c_size, _lib._vp]),
    "gva_aggregate_workspace_bytes": (_lib._c_size, [_lib._c_int] * 4),
    "gva_aggregate_forward_hip_launcher": (_lib._c_int, [_lib._c_int] * 4 + [_lib._vp] * 14 + [_lib._vp]),
    "gva_aggregate_backward_hip_launcher": (_lib._c_int, [_lib._c_int] * 4 + [_lib._vp] * 25 + [_lib._c_size, _lib._vp]),
    "gva_workspace_bytes": (_lib._c_size, [_lib._c_int] * 4),
    "gva_fold_p_forward_hip_launcher": (_lib._c_int, [_lib._c_int] + [_lib._vp] * 9 + [_lib._c_int, ctypes.c_double,
                                                                               ctypes.c_float, ctypes.c_float]
                                        + [_lib._vp] * 4),
    "gva_fold_p_backward_hip_launcher": (_lib._c_int, [_lib._c_int] + [_lib._vp] * 7 + [_lib._c_int] + [_lib._vp] * 7),
    "gva_fold_w_forward_hip_launcher": (_lib._c_int, [_lib._c_int] + [_lib._vp] * 7 + [_lib._c_int, ctypes.c_double,
                                                                               ctypes.c_float, ctypes.c_float]
                                        + [_lib._vp] * 5),
    "gva_fold_w_backward_hip_launcher": (_lib._c_int, [_lib._c_int] + [_lib._vp] * 3 + [_lib._c_int, ctypes.c_double]
                                         + [_lib._vp] * 7),
    "gva_block_workspace_bytes": (_lib._c_size, [_lib._c_int] * 4),
    "gva_block_forward_hip_launcher": (_lib._c_int, [_lib._vp, _lib._vp, _lib._c_size, _lib._vp]),
    "gva_block_backward_hip_launcher": (_lib._c_int, [_lib._vp, _lib._vp, _lib._vp, _lib._c_size, _lib._vp]),
    "gva_peb_forward_hip_launcher": (_lib._c_int, [_lib._c_int] * 3 + [_lib._vp] * 7),
    "gva_peb_backward_hip_launcher": (_lib._c_int, [_lib._c_int] * 3 + [_lib._vp] * 6),
    "gva_attention_forward_hip_launcher": (_lib._c_int, [_lib._c_int] * 4 + [_lib._vp] * 17),
    "gva_attention_backward_hip_launcher": (_lib._c_int, [_lib._c_int] * 4 + [_lib._vp] * 25 + [_lib._c_size, _lib._vp]),
}
_lib.register(_SIG)


# ------------------------------------------------------------------ HIP stages --
class _HipImpl:
    """The product implementation: three launcher families of ao_amd/csrc/gva.hip."""

    @staticmethod
    def pos_stats(coord, idx):
        _lib.require_cuda(coord, idx)
        n, k = idx.shape
        s1 = torch.empty(3, dtype=torch.float64, device=coord.device)
        s2 = torch.empty(9, dtype=torch.float64, device=coord.device)
        L = _lib.lib()
        ws = _lib.workspace(L.gva_workspace_bytes(n, k, 8, 1), coord.device)
        rc = L.gva_pos_stats_hip_launcher(n, k, coord.data_ptr(), idx.data_ptr(), s1.data_ptr(), s2.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "gva_pos_stats_hip_launcher")
        return s1, s2.view(3, 3)

    @staticmethod
    def pos_moments(coord, idx):
        """(mu (3,), cov (3,3)) float64 in two launches (the statistics kernel + one finalize that also forms them)."""
        _lib.require_cuda(coord, idx)
        n, k = idx.shape
        mu = torch.empty(3, dtype=torch.float64, device=coord.device)
        cov = torch.empty(9, dtype=torch.float64, device=coord.device)
        L = _lib.lib()
        ws = _lib.workspace(L.gva_workspace_bytes(n, k, 8, 1), coord.device)
        rc = L.gva_pos_moments_hip_launcher(n, k, coord.data_ptr(), idx.data_ptr(), mu.data_ptr(), cov.data_ptr(),
                                            ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "gva_pos_moments_hip_launcher")
        return mu, cov.view(3, 3)

    @staticmethod
    def logits(kW, qW, a, b, M, cW, coord, idx):
        return _Logits.apply(kW, qW, a, b, M, cW, coord, idx)

    @staticmethod
    def aggregate(W1, sc, sh, Ww2, bw2, v, a, b, coord, idx):
        return _Aggregate.apply(W1, sc, sh, Ww2, bw2, v, a, b, coord, idx)

    @staticmethod
    def project(A, Wp2, bp2, sw, out_v):
        return _PebProject.apply(A, Wp2, bp2, sw, out_v)

    @staticmethod
    def fold_p(lin, bn, mu, cov, rows, training):
        return _FoldP.apply(lin.weight, lin.bias, bn.weight, bn.bias, mu, cov, bn, float(rows), bool(training))

    @staticmethod
    def fold_w(T1, T2, bn, rows, training):
        return _FoldW.apply(T1, T2, bn.weight, bn.bias, bn, float(rows), bool(training))


def _f32c(t):
    return t.detach().float().contiguous()


class _InverseJob(ctypes.Structure):  # mirrors ptv2_inverse_job (include/ptv2_hip.h)
    _fields_ = [("n", ctypes.c_int), ("k", ctypes.c_int), ("idx", ctypes.c_void_p), ("inv_ptr", ctypes.c_void_p),
                ("inv_rows", ctypes.c_void_p)]


INVERSE_MAX_JOBS = 16  # PTV2_INVERSE_MAX_JOBS


def _cached_inverse(idx):
    cached = getattr(idx, "_ao_inverse", None)
    if cached is not None and cached[0] == idx._version:
        return cached[1], cached[2]
    return None


def inverse_tables(tables):
    """inverse_table() of several tables of one device in ONE native call (five launches whatever the number of tables:
    ao_amd/csrc/inverse.hip); results are cached on the tables, tables with a valid cache are skipped."""
    todo = [t for t in tables if t is not None and t.is_cuda and _cached_inverse(t) is None]
    todo = list({id(t): t for t in todo}.values())
    L = _lib.lib()
    for at in range(0, len(todo), INVERSE_MAX_JOBS):
        part = todo[at:at + INVERSE_MAX_JOBS]
        jobs = (_InverseJob * len(part))()
        outs = []
        for job, idx in zip(jobs, part):
            assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.dim() == 2
            n, k = idx.shape
            inv_ptr = torch.empty(n + 1, dtype=torch.int32, device=idx.device)
            inv_rows = torch.empty(n * k, dtype=torch.int32, device=idx.device)
            job.n, job.k, job.idx, job.inv_ptr, job.inv_rows = n, k, idx.data_ptr(), inv_ptr.data_ptr(), inv_rows.data_ptr()
            outs.append((inv_ptr, inv_rows))
        ws = _lib.workspace(L.inverse_tables_hip_workspace_bytes(len(part), ctypes.addressof(jobs)), part[0].device)
        rc = L.inverse_tables_hip_launcher(len(part), ctypes.addressof(jobs), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "inverse_tables_hip_launcher")
        for idx, (inv_ptr, inv_rows) in zip(part, outs):
            try:
                idx._ao_inverse = (idx._version, inv_ptr, inv_rows)
            except AttributeError:
                pass
    return [inverse_table(t) if t is not None else None for t in tables]


def inverse_table(idx):
    """CSR inverse of the neighbour table: for every point j the slots r = n*K + s with idx[r] == j, in
    ascending r (inv_ptr (N+1,) int32, inv_rows int32).  Depends on the table only -> cached on it; the
    backward kernels use it to turn scatter-adds into fixed-order gathers (no float atomics)."""
    cached = _cached_inverse(idx)
    if cached is not None:
        return cached
    n, k = idx.shape
    if idx.is_cuda:
        inv_ptr = torch.empty(n + 1, dtype=torch.int32, device=idx.device)
        inv_rows = torch.empty(n * k, dtype=torch.int32, device=idx.device)
        L = _lib.lib()
        ws = _lib.workspace(L.inverse_table_hip_workspace_bytes(n, k), idx.device)
        rc = L.inverse_table_hip_launcher(n, k, idx.data_ptr(), inv_ptr.data_ptr(), inv_rows.data_ptr(), ws.data_ptr(),
                                          ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "inverse_table_hip_launcher")
    else:  # host statement of the same table (CPU tests of the host logic)
        flat = idx.reshape(-1)
        inv_rows = torch.sort(flat, stable=True)[1].int().contiguous()  # slots ordered by target, -1 first
        counts = torch.bincount((flat + 1).long(), minlength=n + 1)     # bin 0 = the -1 placeholders
        inv_ptr = torch.cumsum(counts, 0).int().contiguous()            # list of j is [inv_ptr[j], inv_ptr[j+1])
    try:
        idx._ao_inverse = (idx._version, inv_ptr, inv_rows)
    except AttributeError:
        pass
    return inv_ptr, inv_rows


class _Logits(torch.autograd.Function):
    """W1 (N,K,G) and its per-channel sums; see module docstring."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, kW, qW, a, b, M, cW, coord, idx):
        _lib.require_cuda(kW, qW, a, b, M, cW, coord, idx)
        kW, qW, a, b, M, cW = (_f32c(t) for t in (kW, qW, a, b, M, cW))
        n, k = idx.shape
        c, g = M.shape
        dev = kW.device
        W1 = torch.empty((n, k, g), dtype=torch.float32, device=dev)
        T1 = torch.empty(g, dtype=torch.float64, device=dev)
        T2 = torch.empty(g, dtype=torch.float64, device=dev)
        L = _lib.lib()
        ws = _lib.workspace(L.gva_workspace_bytes(n, k, c, g), dev)
        rc = L.gva_logits_forward_hip_launcher(
            n, k, c, g, kW.data_ptr(), qW.data_ptr(), a.data_ptr(), b.data_ptr(), M.data_ptr(), cW.data_ptr(),
            coord.data_ptr(), idx.data_ptr(), W1.data_ptr(), T1.data_ptr(), T2.data_ptr(), ws.data_ptr(),
            ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "gva_logits_forward_hip_launcher")
        inv_ptr, inv_rows = inverse_table(idx) if any(ctx.needs_input_grad) else (None, None)
        ctx.save_for_backward(a, b, M, coord, idx, W1, inv_ptr, inv_rows)
        return W1, T1, T2

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gW1, gT1, gT2):
        a, b, M, coord, idx, W1, inv_ptr, inv_rows = ctx.saved_tensors
        n, k = idx.shape
        c, g = M.shape
        dev = W1.device
        gW1 = torch.zeros_like(W1) if gW1 is None else gW1.contiguous()
        gT1 = torch.zeros(g, dtype=torch.float64, device=dev) if gT1 is None else gT1.contiguous()
        gT2 = torch.zeros(g, dtype=torch.float64, device=dev) if gT2 is None else gT2.contiguous()
        gkW = torch.empty((n, g), dtype=torch.float32, device=dev) if inv_ptr is not None else torch.zeros((n, g), dtype=torch.float32, device=dev)
        gqW = torch.empty((n, g), dtype=torch.float32, device=dev)
        ga = torch.empty((c, 3), dtype=torch.float32, device=dev)
        gb = torch.empty(c, dtype=torch.float32, device=dev)
        gM = torch.empty((c, g), dtype=torch.float32, device=dev)
        gcW = torch.empty(g, dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws = _lib.workspace(L.gva_workspace_bytes(n, k, c, g), dev)
        rc = L.gva_logits_backward_hip_launcher(
            n, k, c, g, a.data_ptr(), b.data_ptr(), M.data_ptr(), coord.data_ptr(), idx.data_ptr(),
            W1.data_ptr(), gW1.data_ptr(), gT1.data_ptr(), gT2.data_ptr(), inv_ptr.data_ptr(), inv_rows.data_ptr(),
            gkW.data_ptr(), gqW.data_ptr(),
            ga.data_ptr(), gb.data_ptr(), gM.data_ptr(), gcW.data_ptr(), ws.data_ptr(), ws.numel(),
            _lib.stream_ptr())
        _lib.check(rc, "gva_logits_backward_hip_launcher")
        return gkW, gqW, ga, gb, gM, gcW, None, None


class _Aggregate(torch.autograd.Function):
    """(out_v (N,C), A (N,G,C), sw (N,G)); see module docstring."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx):
        _lib.require_cuda(W1, sc, sh, Ww2, bw2, v, a, b, coord, idx)
        W1, sc, sh, Ww2, bw2, v, a, b = (_f32c(t) for t in (W1, sc, sh, Ww2, bw2, v, a, b))
        n, k = idx.shape
        c = v.shape[1]
        g = sc.shape[0]
        dev = v.device
        out_v = torch.empty((n, c), dtype=torch.float32, device=dev)
        A = torch.empty((n, g, c), dtype=torch.float32, device=dev)
        sw = torch.empty((n, g), dtype=torch.float32, device=dev)
        w = torch.empty((n, k, g), dtype=torch.float32, device=dev)
        rc = _lib.lib().gva_aggregate_forward_hip_launcher(
            n, k, c, g, W1.data_ptr(), sc.data_ptr(), sh.data_ptr(), Ww2.data_ptr(), bw2.data_ptr(), v.data_ptr(),
            a.data_ptr(), b.data_ptr(), coord.data_ptr(), idx.data_ptr(), out_v.data_ptr(), A.data_ptr(),
            sw.data_ptr(), w.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "gva_aggregate_forward_hip_launcher")
        inv_ptr, inv_rows = inverse_table(idx) if any(ctx.needs_input_grad) else (None, None)
        ctx.save_for_backward(W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, inv_ptr, inv_rows, w)
        return out_v, A, sw

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_out, g_A, g_sw):
        W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, inv_ptr, inv_rows, w = ctx.saved_tensors
        n, k = idx.shape
        c = v.shape[1]
        g = sc.shape[0]
        dev = v.device
        g_out = torch.zeros((n, c), dtype=torch.float32, device=dev) if g_out is None else g_out.contiguous()
        g_A = torch.zeros((n, g, c), dtype=torch.float32, device=dev) if g_A is None else g_A.contiguous()
        g_sw = torch.zeros((n, g), dtype=torch.float32, device=dev) if g_sw is None else g_sw.contiguous()
        gW1 = torch.empty_like(W1)
        gv = torch.empty((n, c), dtype=torch.float32, device=dev) if inv_ptr is not None else torch.zeros((n, c), dtype=torch.float32, device=dev)
        gsc = torch.empty(g, dtype=torch.float32, device=dev)
        gsh = torch.empty(g, dtype=torch.float32, device=dev)
        gWw2 = torch.empty((g, g), dtype=torch.float32, device=dev)
        gbw2 = torch.empty(g, dtype=torch.float32, device=dev)
        ga = torch.empty((c, 3), dtype=torch.float32, device=dev)
        gb = torch.empty(c, dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws = _lib.workspace(L.gva_aggregate_workspace_bytes(n, k, c, g), dev)
        rc = L.gva_aggregate_backward_hip_launcher(
            n, k, c, g, W1.data_ptr(), sc.data_ptr(), sh.data_ptr(), Ww2.data_ptr(), bw2.data_ptr(), v.data_ptr(),
            a.data_ptr(), b.data_ptr(), coord.data_ptr(), idx.data_ptr(), w.data_ptr(), g_out.data_ptr(), g_A.data_ptr(),
            g_sw.data_ptr(), inv_ptr.data_ptr(), inv_rows.data_ptr(), gW1.data_ptr(), gsc.data_ptr(), gsh.data_ptr(),
            gWw2.data_ptr(), gbw2.data_ptr(),
            gv.data_ptr(), ga.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "gva_aggregate_backward_hip_launcher")
        return gW1, gsc, gsh, gWw2, gbw2, gv, ga, gb, None, None


def supported(channels, groups, k):
    """Shapes the fused kernels are instantiated for (gva_fwd.hip / gva_bwd.hip); anything else runs the
    unfused composition of gather ops."""
    i = channels // groups
    return (groups in (6, 12, 24, 48, 64) and channels % groups == 0 and i in (2, 4, 8) and channels % 4 == 0
            and k & (k - 1) == 0 and 2 <= k <= 64)


def dropout_supported(channels, groups, k):
    """Shapes for which attention dropout runs inside the fused kernels: the forward softmax kernels take it everywhere,
    the backward needs the fused point kernel (ao_amd/csrc/gva_bwd_point.hip: the five (G, C) instances, k <= 16)."""
    return ((groups, channels) in ((6, 48), (12, 96), (24, 192), (48, 384), (64, 512)) and k & (k - 1) == 0 and 2 <= k <= 16
            and not os.environ.get("AO_AMD_BWD_STAGED"))


def _track(bn, training):
    return training and bn.track_running_stats and bn.running_mean is not None


class _FoldP(torch.autograd.Function):
    """(a (C,3), b (C)) of P = ReLU(pos a^T + b) from linear_p_bias[0] and its BatchNorm (gva_fold.hip)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, Wp1, bp1, gamma, beta, mu, cov, bn, rows, training):
        c = Wp1.shape[0]
        dev = Wp1.device
        a = torch.empty((c, 3), dtype=torch.float32, device=dev)
        b = torch.empty(c, dtype=torch.float32, device=dev)
        rstd = torch.empty(c, dtype=torch.float32, device=dev)
        trk = _track(bn, training)
        if not training and bn.running_mean is None:
            raise RuntimeError("eval-mode fold needs running statistics")
        need_run = trk or not training
        mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
        rc = _lib.lib().gva_fold_p_forward_hip_launcher(
            c, Wp1.data_ptr(), bp1.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _lib.ptr(mu), _lib.ptr(cov),
            bn.running_mean.data_ptr() if need_run else 0, bn.running_var.data_ptr() if need_run else 0,
            bn.num_batches_tracked.data_ptr() if trk else 0, int(training), rows, float(bn.eps), float(mom),
            a.data_ptr(), b.data_ptr(), rstd.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "gva_fold_p_forward_hip_launcher")
        ctx.save_for_backward(Wp1, bp1, gamma, mu, cov, rstd)
        ctx.bn, ctx.training = bn, training
        return a, b

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, ga, gb):
        Wp1, bp1, gamma, mu, cov, rstd = ctx.saved_tensors
        c = Wp1.shape[0]
        gWp1, gbp1 = torch.empty_like(Wp1), torch.empty_like(bp1)
        ggamma, gbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        rm = ctx.bn.running_mean
        rc = _lib.lib().gva_fold_p_backward_hip_launcher(
            c, Wp1.data_ptr(), bp1.data_ptr(), gamma.data_ptr(), _lib.ptr(mu), _lib.ptr(cov),
            rm.data_ptr() if rm is not None else 0, rstd.data_ptr(), int(ctx.training), ga.contiguous().data_ptr(),
            gb.contiguous().data_ptr(), gWp1.data_ptr(), gbp1.data_ptr(), ggamma.data_ptr(), gbeta.data_ptr(),
            _lib.stream_ptr())
        _lib.check(rc, "gva_fold_p_backward_hip_launcher")
        return gWp1, gbp1, ggamma, gbeta, None, None, None, None, None


class _FoldW(torch.autograd.Function):
    """(sc, sh) of the BatchNorm over the logits from their column sums (gva_fold.hip)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, T1, T2, gamma, beta, bn, rows, training):
        g = gamma.shape[0]
        dev = gamma.device
        sc = torch.empty(g, dtype=torch.float32, device=dev)
        sh = torch.empty(g, dtype=torch.float32, device=dev)
        mean = torch.empty(g, dtype=torch.float64, device=dev)
        rstd = torch.empty(g, dtype=torch.float64, device=dev)
        trk = _track(bn, training)
        need_run = trk or not training
        mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
        rc = _lib.lib().gva_fold_w_forward_hip_launcher(
            g, T1.data_ptr(), T2.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
            bn.running_mean.data_ptr() if need_run else 0, bn.running_var.data_ptr() if need_run else 0,
            bn.num_batches_tracked.data_ptr() if trk else 0, int(training), rows, float(bn.eps), float(mom),
            sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "gva_fold_w_forward_hip_launcher")
        ctx.save_for_backward(gamma, mean, rstd)
        ctx.training, ctx.rows = training, rows
        return sc, sh

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gsc, gsh):
        gamma, mean, rstd = ctx.saved_tensors
        g = gamma.shape[0]
        gT1 = torch.empty(g, dtype=torch.float64, device=gamma.device)
        gT2 = torch.empty(g, dtype=torch.float64, device=gamma.device)
        ggamma, gbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        rc = _lib.lib().gva_fold_w_backward_hip_launcher(
            g, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), int(ctx.training), ctx.rows,
            gsc.contiguous().data_ptr(), gsh.contiguous().data_ptr(), gT1.data_ptr(), gT2.data_ptr(), ggamma.data_ptr(),
            gbeta.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "gva_fold_w_backward_hip_launcher")
        return gT1, gT2, ggamma, gbeta, None, None, None


class _PebProject(torch.autograd.Function):
    """out = out_v + (A x Wp2 per group) + bp2 * sw  (gva_peb.hip); grad Wp2 / bp2 are dense rocBLAS products."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, A, Wp2, bp2, sw, out_v):
        _lib.require_cuda(A, Wp2, bp2, sw, out_v)
        A, Wp2, bp2, sw, out_v = (_f32c(t) for t in (A, Wp2, bp2, sw, out_v))
        n, g, c = A.shape
        out = torch.empty((n, c), dtype=torch.float32, device=A.device)
        rc = _lib.lib().gva_peb_forward_hip_launcher(n, c, g, A.data_ptr(), Wp2.data_ptr(), bp2.data_ptr(),
                                                     sw.data_ptr(), out_v.data_ptr(), out.data_ptr(),
                                                     _lib.stream_ptr())
        _lib.check(rc, "gva_peb_forward_hip_launcher")
        ctx.save_for_backward(A, Wp2, bp2, sw)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_out):
        A, Wp2, bp2, sw = ctx.saved_tensors
        n, g, c = A.shape
        i = c // g
        g_out = g_out.contiguous()
        g_A = torch.empty_like(A)
        g_sw = torch.empty_like(sw)
        rc = _lib.lib().gva_peb_backward_hip_launcher(n, c, g, g_out.data_ptr(), Wp2.data_ptr(), bp2.data_ptr(),
                                                      g_A.data_ptr(), g_sw.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "gva_peb_backward_hip_launcher")
        # grad Wp2[g*I+i, c'] = sum_n g_out[n, g*I+i] A[n,g,c']: G batched (I x C') weight-gradient reductions
        g_Wp2 = torch.empty((c, c), dtype=torch.float32, device=A.device)
        L = _lib.lib()
        ws = _lib.workspace(L.dense_workspace_bytes(n, c, c), A.device)
        rc = L.linear_wgrad_strided_hip_launcher(n, i, c, g, g_out.data_ptr(), c, i, A.data_ptr(), g * c, c,
                                                 g_Wp2.data_ptr(), 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "linear_wgrad_strided_hip_launcher")
        g_bp2 = (g_out.view(n, g, i) * sw.unsqueeze(-1)).sum(0).reshape(c)
        return g_A, g_Wp2, g_bp2, g_sw, g_out


# ------------------------------------------------------- whole block, one call --
_P = ctypes.c_void_p


class _BlockArgs(ctypes.Structure):  # mirrors ptv2_gva_block (include/ptv2_hip.h)
    _fields_ = ([(n_, ctypes.c_int) for n_ in ("n", "k", "c", "g", "training")]
                + [(n_, ctypes.c_float) for n_ in ("eps_p", "momentum_p", "eps_w", "momentum_w")]
                + [(n_, _P) for n_ in (
                    "q", "key", "v", "coord", "idx", "mu", "cov",
                    "Wp1", "bp1", "gamma_p", "beta_p", "Wp2", "bp2", "Ww1", "bw1", "gamma_w", "beta_w", "Ww2", "bw2",
                    "run_mean_p", "run_var_p", "run_mean_w", "run_var_w", "batches_p", "batches_w",
                    "out", "a", "b", "rstd_p", "M", "cW", "kW", "qW", "W1", "w", "A", "sw", "sc", "sh",
                    "mean_w", "rstd_w", "q_sc", "q_sh", "k_sc", "k_sh")]
                + [("attn_drop_p", ctypes.c_float), ("attn_drop_seed", ctypes.c_uint)])


_lib.check_struct(3, _BlockArgs)  # (the mirror must have the size the library was compiled with)


class _BlockGrads(ctypes.Structure):  # mirrors ptv2_gva_block_grads
    _fields_ = [(n_, _P) for n_ in (
        "g_out", "inv_ptr", "inv_rows", "gq", "gk", "gv", "gWp1", "gbp1", "ggamma_p", "gbeta_p", "gWp2", "gbp2",
        "gWw1", "gbw1", "ggamma_w", "gbeta_w", "gWw2", "gbw2")]


def _momentum(bn):
    return bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)


class _GvaBlock(torch.autograd.Function):
    """GroupedVectorAttention core (everything after linear_q/k/v) as one native call per direction
    (ao_amd/csrc/gva_block.hip)."""

    PARAMS = ("Wp1", "bp1", "gamma_p", "beta_p", "Wp2", "bp2", "Ww1", "bw1", "gamma_w", "beta_w", "Ww2", "bw2")

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, q, key, v, coord, idx, mu, cov, bn_p, bn_w, training, drop, *params):
        q, key, v = (_f32c(t) for t in (q, key, v))
        params = [p.contiguous() for p in params]
        n, k = idx.shape
        c, g = q.shape[1], params[6].shape[0]
        dev = q.device
        f32 = dict(dtype=torch.float32, device=dev)
        f64 = dict(dtype=torch.float64, device=dev)
        sv = dict(a=torch.empty((c, 3), **f32), b=torch.empty(c, **f32), rstd_p=torch.empty(c, **f32),
                  M=torch.empty((c, g), **f32), cW=torch.empty(g, **f32), kW=torch.empty((n, g), **f32),
                  qW=torch.empty((n, g), **f32), W1=torch.empty((n, k, g), **f32), w=torch.empty((n, k, g), **f32),
                  A=torch.empty((n, g, c), **f32), sw=torch.empty((n, g), **f32), sc=torch.empty(g, **f32),
                  sh=torch.empty(g, **f32), mean_w=torch.empty(g, **f64), rstd_w=torch.empty(g, **f64))
        out = torch.empty((n, c), **f32)
        args = _BlockArgs()
        args.n, args.k, args.c, args.g, args.training = n, k, c, g, int(training)
        args.eps_p, args.momentum_p = float(bn_p.eps), float(_momentum(bn_p))
        args.eps_w, args.momentum_w = float(bn_w.eps), float(_momentum(bn_w))
        for name, t in (("q", q), ("key", key), ("v", v), ("coord", coord), ("idx", idx), ("mu", mu), ("cov", cov),
                        ("out", out)):
            setattr(args, name, _lib.ptr(t))
        for name, t in zip(_GvaBlock.PARAMS, params):
            setattr(args, name, t.data_ptr())
        for tag, bn in (("p", bn_p), ("w", bn_w)):
            trk = _track(bn, training)
            need = trk or not training
            setattr(args, "run_mean_" + tag, bn.running_mean.data_ptr() if need else 0)
            setattr(args, "run_var_" + tag, bn.running_var.data_ptr() if need else 0)
            setattr(args, "batches_" + tag, bn.num_batches_tracked.data_ptr() if trk else 0)
        for name, t in sv.items():
            setattr(args, name, t.data_ptr())
        args.attn_drop_p, args.attn_drop_seed = (float(drop[0]), int(drop[1])) if (drop and training) else (0.0, 0)
        ctx.drop = (args.attn_drop_p, args.attn_drop_seed)
        L = _lib.lib()
        ws = _lib.workspace(L.gva_block_workspace_bytes(n, k, c, g), dev)
        rc = L.gva_block_forward_hip_launcher(ctypes.addressof(args), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "gva_block_forward_hip_launcher")
        need_grad = any(ctx.needs_input_grad)
        inv_ptr, inv_rows = inverse_table(idx) if need_grad else (None, None)
        ctx.save_for_backward(q, key, v, coord, idx, mu, cov, inv_ptr, inv_rows, *params, *sv.values())
        ctx.sv_names = list(sv.keys())
        ctx.bns, ctx.training, ctx.dims = (bn_p, bn_w), training, (n, k, c, g)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_out):
        saved = ctx.saved_tensors
        q, key, v, coord, idx, mu, cov, inv_ptr, inv_rows = saved[:9]
        params = saved[9:9 + len(_GvaBlock.PARAMS)]
        sv = dict(zip(ctx.sv_names, saved[9 + len(_GvaBlock.PARAMS):]))
        n, k, c, g = ctx.dims
        bn_p, bn_w = ctx.bns
        dev = q.device
        g_out = g_out.contiguous()
        args = _BlockArgs()
        args.n, args.k, args.c, args.g, args.training = n, k, c, g, int(ctx.training)
        args.eps_p, args.momentum_p, args.eps_w, args.momentum_w = float(bn_p.eps), 0.0, float(bn_w.eps), 0.0
        for name, t in (("q", q), ("key", key), ("v", v), ("coord", coord), ("idx", idx), ("mu", mu), ("cov", cov)):
            setattr(args, name, _lib.ptr(t))
        for name, t in zip(_GvaBlock.PARAMS, params):
            setattr(args, name, t.data_ptr())
        args.run_mean_p = bn_p.running_mean.data_ptr() if bn_p.running_mean is not None else 0
        args.run_mean_w = bn_w.running_mean.data_ptr() if bn_w.running_mean is not None else 0
        for name, t in sv.items():
            setattr(args, name, t.data_ptr())
        args.attn_drop_p, args.attn_drop_seed = ctx.drop  # the forward's mask again
        grads = _BlockGrads()
        gq, gk = torch.empty_like(q), torch.empty_like(key)
        gv = torch.empty_like(v) if inv_ptr is not None else torch.zeros_like(v)
        gp = [torch.empty_like(p) for p in params]
        grads.g_out, grads.inv_ptr, grads.inv_rows = g_out.data_ptr(), _lib.ptr(inv_ptr), _lib.ptr(inv_rows)
        grads.gq, grads.gk, grads.gv = gq.data_ptr(), gk.data_ptr(), gv.data_ptr()
        for name, t in zip(("gWp1", "gbp1", "ggamma_p", "gbeta_p", "gWp2", "gbp2", "gWw1", "gbw1", "ggamma_w", "gbeta_w",
                            "gWw2", "gbw2"), gp):
            setattr(grads, name, t.data_ptr())
        L = _lib.lib()
        ws = _lib.workspace(L.gva_block_workspace_bytes(n, k, c, g), dev)
        rc = L.gva_block_backward_hip_launcher(ctypes.addressof(args), ctypes.addressof(grads), ws.data_ptr(), ws.numel(),
                                               _lib.stream_ptr())
        _lib.check(rc, "gva_block_backward_hip_launcher")
        return (gq, gk, gv, None, None, None, None, None, None, None, None, *gp)


def _block_call(mod, query, key, value, coord, idx):
    lin_p1, bn_p, lin_p2 = mod.linear_p_bias[0], mod.linear_p_bias[1].norm, mod.linear_p_bias[3]
    lin_w1, bn_w, lin_w2 = mod.weight_encoding[0], mod.weight_encoding[1].norm, mod.weight_encoding[3]
    training = mod.training or not bn_p.track_running_stats or bn_p.running_mean is None
    mu = cov = None
    if training:
        mu, cov = _pos_moments(_HipImpl, coord, idx)
    drop = (mod.attn_drop_rate, next_drop_seed()) if (mod.training and mod.attn_drop_rate > 0.0) else None
    return _GvaBlock.apply(query, key, value, coord, idx, mu, cov, bn_p, bn_w, training, drop,
                           lin_p1.weight, lin_p1.bias, bn_p.weight, bn_p.bias, lin_p2.weight, lin_p2.bias,
                           lin_w1.weight, lin_w1.bias, bn_w.weight, bn_w.bias, lin_w2.weight, lin_w2.bias)


# ---------------------------------------------------------------- attention dropout --
_M32 = 0xFFFFFFFF


def next_drop_seed():
    """A fresh 32-bit seed for one Block's attention-dropout mask of one step, drawn on the HOST from torch's CPU generator
    (reproducible under torch.manual_seed, no device synchronisation)."""
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


def attn_drop_mask(seed, n, k, g, p, device):
    """The dropout factor of every softmax weight, (n, k, g) fp32: 0 or 1 / (1 - p) -- the torch statement of
    ptv2_drop_factor (ao_amd/csrc/gva_common.h), which the fused kernels evaluate in the forward and again in the backward
    instead of storing a mask.  Element e = (point * k + slot) * g + group; a 32-bit integer hash of (e, seed) against
    p * 2^32.  nn.Dropout(p) in the reference (point_transformer_v2m2_base.py:101,122) is the same distribution from
    another generator."""
    if not p > 0.0:
        return torch.ones((n, k, g), dtype=torch.float32, device=device)
    pp = min(float(p), 1.0)
    thresh = min(int(pp * 4294967296.0), _M32)
    thresh = max(thresh, 1)
    scale = 1.0 / (1.0 - pp) if pp < 1.0 else 0.0
    e = torch.arange(n * k * g, dtype=torch.int64, device=device)
    h = (e & _M32) ^ (((e >> 32) * 0x27D4EB2F) & _M32)
    h = (h * 0x9E3779B1 + int(seed)) & _M32
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & _M32
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & _M32
    h = h ^ (h >> 16)
    return ((h >= thresh).to(torch.float32) * scale).view(n, k, g)


# -------------------------------------------------------------------- host logic --
def _pos_moments(impl, coord, idx):
    """(mu (3,), cov (3,3)) of the masked relative positions, float64; cached on the idx tensor because
    every block of a BlockSequence (and the decoder stage at the same resolution) shares the table."""
    cached = getattr(idx, "_ao_pos_moments", None)
    if cached is not None and cached[0] == (coord.data_ptr(), idx._version):
        return cached[1], cached[2]
    with torch.no_grad():
        if hasattr(impl, "pos_moments"):
            mu, cov = impl.pos_moments(coord, idx)
        else:
            s1, s2 = impl.pos_stats(coord, idx)
            rows = idx.shape[0] * idx.shape[1]
            mu = s1 / rows
            cov = s2 / rows - torch.outer(mu, mu)
    try:
        idx._ao_pos_moments = ((coord.data_ptr(), idx._version), mu, cov)
    except AttributeError:
        pass
    return mu, cov


def _bn_update(bn, mean, var_biased, rows):
    """Running-statistics update of nn.BatchNorm1d in training mode (momentum form)."""
    if not bn.track_running_stats or bn.running_mean is None:
        return
    with torch.no_grad():
        bn.num_batches_tracked += 1
        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        unbiased = var_biased * (rows / max(rows - 1, 1))
        bn.running_mean.mul_(1 - m).add_(mean.detach().to(bn.running_mean.dtype), alpha=m)
        bn.running_var.mul_(1 - m).add_(unbiased.detach().to(bn.running_var.dtype), alpha=m)


def grouped_vector_attention(mod, query, key, value, coord, reference_index, impl=None):
    """mod: GroupedVectorAttention (pe_bias=True, pe_multiplier=False); query/key/value (N,C) are the outputs
    of mod.linear_q / linear_k / linear_v.  Returns (N,C).  `impl` swaps the three device stages
    (tests pass a torch restatement to check this host logic on CPU)."""
    if impl is None and os.environ.get("AO_AMD_GVA", "fused") != "staged":
        return _block_call(mod, query.float(), key.float(), value.float(), coord.contiguous(),
                           reference_index.contiguous())
    impl = impl or _HipImpl
    C, G = mod.embed_channels, mod.groups
    I = C // G
    idx = reference_index.contiguous()
    N, K = idx.shape
    rows = N * K
    coord = coord.contiguous()
    query, key, value = query.float(), key.float(), value.float()
    lin_p1, bn_p, lin_p2 = mod.linear_p_bias[0], mod.linear_p_bias[1].norm, mod.linear_p_bias[3]
    lin_w1, bn_w, lin_w2 = mod.weight_encoding[0], mod.weight_encoding[1].norm, mod.weight_encoding[3]
    training_stats = mod.training or not bn_p.track_running_stats or bn_p.running_mean is None

    # -- BN_p folded into an affine map of pos: P = ReLU(pos a^T + b) -----------------------------
    Wp1, bp1 = lin_p1.weight.float(), lin_p1.bias.float()
    mu = cov = None
    if training_stats:
        mu, cov = _pos_moments(impl, coord, idx)
    if hasattr(impl, "fold_p"):
        a, b = impl.fold_p(lin_p1, bn_p, mu, cov, rows, training_stats)
    else:
        if training_stats:
            mu32, cov32 = mu.float(), cov.float()
            mean_p = Wp1 @ mu32 + bp1
            var_p = ((Wp1 @ cov32) * Wp1).sum(1).clamp_min(0)
            if mod.training:
                _bn_update(bn_p, mean_p, var_p, rows)
        else:
            mean_p, var_p = bn_p.running_mean.float(), bn_p.running_var.float()
        scale_p = bn_p.weight.float() * torch.rsqrt(var_p + bn_p.eps)
        a = Wp1 * scale_p.unsqueeze(1)
        b = (bp1 - mean_p) * scale_p + bn_p.bias.float()

    # -- logits: W1 = kW[idx] - qW + P M + cW ------------------------------------------------------
    Wp2, bp2 = lin_p2.weight.float(), lin_p2.bias.float()
    Ww1, bw1 = lin_w1.weight.float(), lin_w1.bias.float()
    M = Wp2.t() @ Ww1.t()                      # (C,G): M[c',g] = sum_c Wp2[c,c'] Ww1[g,c]
    cW = torch.addmv(bw1, Ww1, bp2)
    kW = skinny_linear(key, Ww1)
    qW = skinny_linear(query, Ww1)
    W1, T1, T2 = impl.logits(kW, qW, a, b, M, cW, coord, idx)

    # -- BN_w from the per-channel sums of W1 --------------------------------------------------------
    train_w = mod.training or not bn_w.track_running_stats
    if hasattr(impl, "fold_w"):
        sc, sh = impl.fold_w(T1, T2, bn_w, rows, train_w)
    else:
        if train_w:
            mean_w = T1 / rows
            var_w = (T2 / rows - mean_w * mean_w).clamp_min(0)
            if mod.training:
                _bn_update(bn_w, mean_w, var_w, rows)
        else:
            mean_w, var_w = bn_w.running_mean.double(), bn_w.running_var.double()
        sc64 = bn_w.weight.double() * torch.rsqrt(var_w + bn_w.eps)
        sh64 = bn_w.bias.double() - mean_w * sc64
        sc, sh = sc64.float(), sh64.float()

    # -- softmax over neighbours, aggregation of v and of the folded positional bias ---------------
    out_v, A, sw = impl.aggregate(W1, sc, sh, lin_w2.weight.float(), lin_w2.bias.float(), value, a, b, coord, idx)
    if hasattr(impl, "project"):
        return impl.project(A, Wp2, bp2, sw, out_v)
    peb = torch.einsum("ngc,gic->ngi", A, Wp2.view(G, I, C))
    return out_v + peb.reshape(N, C) + (sw.unsqueeze(-1) * bp2.view(1, G, I)).reshape(N, C)
