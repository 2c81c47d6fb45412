"""Per-point dense layers of PT-v2m2 on the HIP kernels of ao_amd/csrc/dense.hip.

`RowLinear` / `RowBatchNorm1d` subclass nn.Linear / nn.BatchNorm1d, so parameter names, buffers and
state_dict layout are exactly the reference's; only the execution differs:
  * BatchNorm (+ fused ReLU) forward/backward run as HBM-streaming HIP passes,
  * the Linear forward and input gradient stay on rocBLAS (MFMA), the weight/bias gradient is a
    split-K HIP reduction (dW = dY^T X has a tiny output and K = N up to 1e5+).
Anything the kernels do not cover (CPU tensors, c % 4 != 0, non-fp32, cumulative-average momentum)
takes the stock torch path of the parent class.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

WGRAD_MIN_ROWS = 2048


def _ws(n, cout, cin, device):
    return _lib.workspace(_lib.lib().dense_workspace_bytes(n, cout, cin), device)


class _BNRows(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, gamma, beta, bn, relu):
        n, c = x.shape
        L = _lib.lib()
        training = bn.training or bn.running_mean is None
        if training:
            mean = torch.empty(c, dtype=torch.float32, device=x.device)
            rstd = torch.empty(c, dtype=torch.float32, device=x.device)
            track = bn.track_running_stats and bn.training and bn.running_mean is not None
            ws = _ws(n, c, c, x.device)
            rc = L.bn_stats_hip_launcher(
                n, c, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                bn.running_mean.data_ptr() if track else 0, bn.running_var.data_ptr() if track else 0,
                bn.num_batches_tracked.data_ptr() if track else 0, float(bn.eps), float(bn.momentum),
                ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, "bn_stats_hip_launcher")
        else:
            mean = bn.running_mean
            rstd = torch.rsqrt(bn.running_var + bn.eps)
        y = torch.empty_like(x)
        rc = L.bn_apply_hip_launcher(n, c, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                     beta.data_ptr(), int(relu), y.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "bn_apply_hip_launcher")
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.relu, ctx.training = bool(relu), bool(training)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        n, c = x.shape
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        ws = _ws(n, c, c, x.device)
        rc = _lib.lib().bn_backward_hip_launcher(
            n, c, x.data_ptr(), gy.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
            int(ctx.relu), int(ctx.training), gx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(),
            ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "bn_backward_hip_launcher")
        return gx, dgamma, dbeta, None, None


class _BNResidualRelu(torch.autograd.Function):
    """y = ReLU(identity + rowscale * BN(x)): the tail of a Block in one apply pass (dense.hip)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, gamma, beta, identity, rowscale, bn):
        n, c = x.shape
        L = _lib.lib()
        training = bn.training or bn.running_mean is None
        if training:
            mean = torch.empty(c, dtype=torch.float32, device=x.device)
            rstd = torch.empty(c, dtype=torch.float32, device=x.device)
            track = bn.track_running_stats and bn.training and bn.running_mean is not None
            ws = _ws(n, c, c, x.device)
            rc = L.bn_stats_hip_launcher(
                n, c, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                bn.running_mean.data_ptr() if track else 0, bn.running_var.data_ptr() if track else 0,
                bn.num_batches_tracked.data_ptr() if track else 0, float(bn.eps), float(bn.momentum),
                ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, "bn_stats_hip_launcher")
        else:
            mean = bn.running_mean
            rstd = torch.rsqrt(bn.running_var + bn.eps)
        y = torch.empty_like(x)
        rc = L.bn_apply_residual_hip_launcher(n, c, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                              beta.data_ptr(), identity.data_ptr(), _lib.ptr(rowscale), y.data_ptr(),
                                              _lib.stream_ptr())
        _lib.check(rc, "bn_apply_residual_hip_launcher")
        ctx.save_for_backward(x, mean, rstd, gamma, y, rowscale)
        ctx.training = bool(training)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, mean, rstd, gamma, y, rowscale = ctx.saved_tensors
        n, c = x.shape
        gy = gy.contiguous()
        gx, gres = torch.empty_like(x), torch.empty_like(x)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        ws = _ws(n, c, c, x.device)
        rc = _lib.lib().bn_backward_residual_hip_launcher(
            n, c, x.data_ptr(), gy.data_ptr(), y.data_ptr(), _lib.ptr(rowscale), mean.data_ptr(), rstd.data_ptr(),
            gamma.data_ptr(), int(ctx.training), gx.data_ptr(), gres.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
            ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "bn_backward_residual_hip_launcher")
        return gx, dgamma, dbeta, gres, None, None


def bn_residual_relu(bn, x, identity, rowscale=None):
    """ReLU(identity + rowscale[:,None] * bn(x)) with the fused kernels when they apply."""
    if isinstance(bn, RowBatchNorm1d) and bn._fast(x) and identity.dtype == torch.float32:
        return _BNResidualRelu.apply(x.contiguous(), bn.weight, bn.bias, identity.contiguous(),
                                     None if rowscale is None else rowscale.contiguous(), bn)
    y = bn(x)
    if rowscale is not None:
        y = y * rowscale.unsqueeze(1)
    return F.relu(identity + y)


class RowBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d on (N,C) rows; `forward(x, relu=True)` fuses the activation."""

    def _fast(self, x):
        return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 4 == 0
                and 4 <= x.shape[1] <= 1024 and x.shape[0] > 1 and self.affine and self.momentum is not None)

    def forward(self, x, relu=False):
        if not self._fast(x):
            y = super().forward(x)
            return F.relu(y) if relu else y
        return _BNRows.apply(x.contiguous(), self.weight, self.bias, self, relu)


class _LinearRows(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        n, cin = x.shape
        cout = weight.shape[0]
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        ws = _ws(n, cout, cin, x.device)
        rc = _lib.lib().linear_wgrad_hip_launcher(n, cout, cin, gy.data_ptr(), x.data_ptr(), dW.data_ptr(),
                                                  db.data_ptr() if db is not None else 0, ws.data_ptr(), ws.numel(),
                                                  _lib.stream_ptr())
        _lib.check(rc, "linear_wgrad_hip_launcher")
        return gx, dW, db


class RowLinear(nn.Linear):
    """nn.Linear on (N,Cin) rows with the split-K weight gradient."""

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.weight.dtype == torch.float32
                and x.shape[0] >= WGRAD_MIN_ROWS and torch.is_grad_enabled() and self.weight.requires_grad):
            return _LinearRows.apply(x.contiguous(), self.weight, self.bias)
        return super().forward(x)


class _SkinnyLinear(torch.autograd.Function):
    """y = x W^T for a narrow output (cout <= 64): HIP forward / input-gradient kernels, MFMA split-K wgrad."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight):
        x, weight = x.contiguous(), weight.contiguous()
        n, cin = x.shape
        cout = weight.shape[0]
        y = torch.empty((n, cout), dtype=torch.float32, device=x.device)
        rc = _lib.lib().skinny_linear_forward_hip_launcher(n, cin, cout, x.data_ptr(), weight.data_ptr(), y.data_ptr(),
                                                           _lib.stream_ptr())
        _lib.check(rc, "skinny_linear_forward_hip_launcher")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        n, cin = x.shape
        cout = weight.shape[0]
        L = _lib.lib()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            rc = L.skinny_linear_backward_hip_launcher(n, cin, cout, gy.data_ptr(), weight.data_ptr(), gx.data_ptr(),
                                                       _lib.stream_ptr())
            _lib.check(rc, "skinny_linear_backward_hip_launcher")
        dW = torch.empty_like(weight)
        ws = _ws(n, cout, cin, x.device)
        rc = L.linear_wgrad_hip_launcher(n, cout, cin, gy.data_ptr(), x.data_ptr(), dW.data_ptr(), 0, ws.data_ptr(),
                                         ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "linear_wgrad_hip_launcher")
        return gx, dW


def skinny_linear(x, weight):
    """x (N,Cin) @ weight (Cout,Cin)^T with Cout <= 64; falls back to torch off-GPU / odd shapes."""
    if (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2
            and x.shape[1] % 4 == 0 and weight.shape[0] <= 64 and x.shape[0] >= 256):
        return _SkinnyLinear.apply(x, weight)
    return x @ weight.t()


class _LinBnRelu(torch.autograd.Function):
    """ReLU(BatchNorm(x W^T + b)) as one autograd node on the library's row GEMM / BatchNorm / weight-gradient
    kernels (the per-stage projections around the Blocks: GridPool.fc, Unpool.proj / proj_skip, the head)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias, gamma, beta, bn):
        x = x.contiguous()
        n, cin = x.shape
        cout = weight.shape[0]
        L = _lib.lib()
        dev = x.device
        st = _lib.stream_ptr()
        h = torch.empty((n, cout), dtype=torch.float32, device=dev)
        rc = L.rows_gemm_hip_launcher(n, cout, cin, x.data_ptr(), weight.data_ptr(), 0, _lib.ptr(bias), h.data_ptr(), 0, st)
        _lib.check(rc, "rows_gemm_hip_launcher")
        training = bn.training or bn.running_mean is None
        y = torch.empty_like(h)
        if training:
            mean = torch.empty(cout, dtype=torch.float32, device=dev)
            rstd = torch.empty(cout, dtype=torch.float32, device=dev)
            track = bn.track_running_stats and bn.training and bn.running_mean is not None
            ws = _ws(n, cout, cout, dev)
            rc = L.bn_forward_hip_launcher(
                n, cout, h.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, mean.data_ptr(), rstd.data_ptr(),
                bn.running_mean.data_ptr() if track else 0, bn.running_var.data_ptr() if track else 0,
                bn.num_batches_tracked.data_ptr() if track else 0, float(bn.eps), float(bn.momentum), 0, 0, y.data_ptr(),
                ws.data_ptr(), ws.numel(), st)
            _lib.check(rc, "bn_forward_hip_launcher")
        else:
            mean = bn.running_mean
            rstd = torch.rsqrt(bn.running_var + bn.eps)
            rc = L.bn_apply_hip_launcher(n, cout, h.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                         beta.data_ptr(), 1, y.data_ptr(), st)
            _lib.check(rc, "bn_apply_hip_launcher")
        ctx.save_for_backward(x, weight, h, mean, rstd, gamma, beta)
        ctx.training, ctx.has_bias = bool(training), bias is not None
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, weight, h, mean, rstd, gamma, beta = ctx.saved_tensors
        n, cin = x.shape
        cout = weight.shape[0]
        L = _lib.lib()
        dev = x.device
        st = _lib.stream_ptr()
        gy = gy.contiguous()
        gh = torch.empty_like(h)
        dgamma = torch.empty(cout, dtype=torch.float32, device=dev)
        dbeta = torch.empty(cout, dtype=torch.float32, device=dev)
        ws = _ws(n, max(cout, cin), max(cout, cin), dev)
        rc = L.bn_backward_hip_launcher(n, cout, h.data_ptr(), gy.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                        beta.data_ptr(), 1, int(ctx.training), gh.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                        ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "bn_backward_hip_launcher")
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            rc = L.rows_gemm_hip_launcher(n, cin, cout, gh.data_ptr(), weight.data_ptr(), 1, 0, gx.data_ptr(), 0, st)
            _lib.check(rc, "rows_gemm_hip_launcher")
        dW = torch.empty_like(weight)
        db = torch.empty(cout, dtype=torch.float32, device=dev) if ctx.has_bias else None
        rc = L.linear_wgrad_hip_launcher(n, cout, cin, gh.data_ptr(), x.data_ptr(), dW.data_ptr(), _lib.ptr(db), ws.data_ptr(),
                                         ws.numel(), st)
        _lib.check(rc, "linear_wgrad_hip_launcher")
        return gx, dW, db, dgamma, dbeta, None


def lin_bn_relu(linear, bn, x):
    """ReLU(bn(linear(x))) for nn.Linear `linear` and RowBatchNorm1d `bn`; one fused autograd node when the row
    kernels cover the shape (fp32 CUDA rows, channel counts multiples of 4), the two-module path otherwise."""
    cout, cin = linear.weight.shape
    if (isinstance(bn, RowBatchNorm1d) and x.is_cuda and x.dim() == 2 and cin % 4 == 0 and cout % 4 == 0
            and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and x.shape[0] > 1 and bn.affine
            and bn.momentum is not None and 4 <= cout <= 1024 and linear.weight.dtype == torch.float32
            and linear.weight.is_contiguous()):
        from . import block  # noqa: F401  (registers rows_gemm_hip_launcher)

        return _LinBnRelu.apply(x, linear.weight, linear.bias, bn.weight, bn.bias, bn)  # fp32 also under autocast
    return bn(linear(x), relu=True)
