"""GPU: RowBatchNorm1d / RowLinear (ao_amd/csrc/dense.hip) against stock torch modules with the same
parameters, forward, backward and running statistics."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


@pytest.mark.parametrize("n,c", [(120000, 48), (18905, 96), (4501, 192), (1074, 384), (37, 512), (5000, 8)])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("training", [True, False])
def test_row_batchnorm(n, c, relu, training):
    from ao_amd.ptv2.layers import RowBatchNorm1d

    torch.manual_seed(0)
    x = (torch.randn(n, c, device="cuda") * 2 + 0.7).requires_grad_(True)
    ref = nn.BatchNorm1d(c).cuda()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.normal_(0, 0.3)
        ref.running_mean.normal_(0.5, 0.2)
        ref.running_var.uniform_(2, 5)
    mine = RowBatchNorm1d(c).cuda()
    mine.load_state_dict(copy.deepcopy(ref.state_dict()))
    ref.train(training)
    mine.train(training)
    y_ref = ref(x)
    y_ref = F.relu(y_ref) if relu else y_ref
    y = mine(x, relu)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().cpu().numpy(), rtol=1e-4, atol=2e-5)
    go = torch.randn_like(y)
    g_ref = torch.autograd.grad(y_ref, [x, ref.weight, ref.bias], go)
    g = torch.autograd.grad(y, [x, mine.weight, mine.bias], go)
    for a, b, nm in zip(g, g_ref, ("gx", "dgamma", "dbeta")):
        assert rel(a, b) < (1e-3 if relu else 2e-4), (nm, rel(a, b))  # relu: a 1-ulp y difference flips a mask bit
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        np.testing.assert_allclose(mine.state_dict()[k].cpu().numpy(), ref.state_dict()[k].cpu().numpy(), rtol=1e-5,
                                   atol=1e-6, err_msg=k)


@pytest.mark.parametrize("n,cin,cout,bias", [(120000, 48, 48, True), (120000, 6, 48, False), (18905, 96, 96, True),
                                              (4501, 192, 96, True), (3000, 384, 384, False), (120000, 48, 13, True),
                                              (2500, 512, 384, True)])
def test_row_linear(n, cin, cout, bias):
    from ao_amd.ptv2.layers import RowLinear

    torch.manual_seed(1)
    x = torch.randn(n, cin, device="cuda", requires_grad=True)
    mine = RowLinear(cin, cout, bias=bias).cuda()
    y = mine(x)
    y_ref = F.linear(x, mine.weight, mine.bias)
    assert torch.equal(y, y_ref)
    go = torch.randn_like(y)
    params = [x, mine.weight] + ([mine.bias] if bias else [])
    g = torch.autograd.grad(y, params, go)
    g_ref = torch.autograd.grad(y_ref, params, go)
    for a, b in zip(g, g_ref):
        assert rel(a, b) < 1e-4, rel(a, b)


@pytest.mark.parametrize("n,cin,cout", [(120000, 48, 6), (18905, 96, 12), (4501, 192, 24), (1074, 384, 48), (300, 512, 64)])
def test_skinny_linear(n, cin, cout):
    from ao_amd.ptv2.layers import skinny_linear

    torch.manual_seed(2)
    x = torch.randn(n, cin, device="cuda", requires_grad=True)
    w = (torch.randn(cout, cin, device="cuda") * 0.2).requires_grad_(True)
    y = skinny_linear(x, w)
    y_ref = x @ w.t()
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    go = torch.randn_like(y)
    g = torch.autograd.grad(y, [x, w], go)
    g_ref = torch.autograd.grad(y_ref, [x, w], go)
    for a, b in zip(g, g_ref):
        assert rel(a, b) < 1e-4, rel(a, b)


@pytest.mark.parametrize("n,c", [(120000, 48), (4501, 192), (300, 384)])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("drop", [False, True])
def test_bn_residual_relu(n, c, training, drop):
    from ao_amd.ptv2.layers import RowBatchNorm1d, bn_residual_relu

    torch.manual_seed(3)
    x = torch.randn(n, c, device="cuda", requires_grad=True)
    ident = torch.randn(n, c, device="cuda", requires_grad=True)
    rowscale = (torch.rand(n, device="cuda") < 0.7).float() / 0.7 if drop else None
    ref = nn.BatchNorm1d(c).cuda()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.normal_(0, 0.3)
    mine = RowBatchNorm1d(c).cuda()
    mine.load_state_dict(copy.deepcopy(ref.state_dict()))
    ref.train(training)
    mine.train(training)
    yr = ref(x)
    yr = F.relu(ident + (yr * rowscale.unsqueeze(1) if drop else yr))
    y = bn_residual_relu(mine, x, ident, rowscale)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-4, atol=2e-5)
    go = torch.randn_like(y)
    g_ref = torch.autograd.grad(yr, [x, ident, ref.weight, ref.bias], go)
    g = torch.autograd.grad(y, [x, ident, mine.weight, mine.bias], go)
    for a, b, nm in zip(g, g_ref, ("gx", "gident", "dgamma", "dbeta")):
        assert rel(a, b) < 1e-3, (nm, rel(a, b))


@pytest.mark.parametrize("n,c,count", [(18905, 96, 1), (4501, 192, 3), (1074, 384, 3), (37, 48, 1), (30001, 48, 3)])
@pytest.mark.parametrize("training", [True, False])
def test_gemm_epilogue_bn_backward_records(n, c, count, training):
    """rows_gemm_bnbwd + bn_backward_records (the reduce pass of a BatchNorm + ReLU backward inside the GEMM that produces
    its incoming gradient) against torch autograd of  relu(bn(x))  fed with  gy = sum_i X_i W_i."""
    from ao_amd import _lib

    L = _lib.lib()
    torch.manual_seed(n + c)
    dev = "cuda"
    X = [torch.randn(n, c, device=dev) for _ in range(count)]
    W = [torch.randn(c, c, device=dev) / c ** 0.5 for _ in range(count)]
    x = (torch.randn(n, c, device=dev) * 1.5 + 0.3).requires_grad_(True)
    bn = nn.BatchNorm1d(c).to(dev).train(training)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0.3, 0.2)
        bn.running_var.uniform_(1.5, 3)
    gy_ref = sum(a @ w for a, w in zip(X, W))
    if training:
        mean = x.detach().mean(0)
        rstd = (x.detach().var(0, unbiased=False) + bn.eps).rsqrt()
    else:
        mean, rstd = bn.running_mean.clone(), (bn.running_var + bn.eps).rsqrt()
    F.relu(bn(x)).backward(gy_ref)

    gy = torch.empty(n, c, device=dev)
    nrec = (n + 63) // 64
    rec = torch.full((nrec * 2 * c + 64,), float("nan"), device=dev)
    gx, dg, db = torch.empty(n, c, device=dev), torch.empty(c, device=dev), torch.empty(c, device=dev)
    import ctypes
    arr = ctypes.c_void_p * count
    xs, ws = arr(*[t.data_ptr() for t in X]), arr(*[t.data_ptr() for t in W])
    g, b = bn.weight.detach(), bn.bias.detach()
    _lib.check(L.rows_gemm_bnbwd_hip_launcher(n, c, c, count, xs, ws, 1, gy.data_ptr(), x.data_ptr(), mean.data_ptr(),
                                              rstd.data_ptr(), g.data_ptr(), b.data_ptr(), 1, rec.data_ptr(),
                                              _lib.stream_ptr()), "rows_gemm_bnbwd")
    _lib.check(L.bn_backward_records_hip_launcher(n, c, x.data_ptr(), gy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                  g.data_ptr(), b.data_ptr(), 1, int(training), gx.data_ptr(), dg.data_ptr(),
                                                  db.data_ptr(), rec.data_ptr(), nrec, _lib.stream_ptr()), "bn_backward_records")
    torch.cuda.synchronize()
    assert rel(gy, gy_ref) < 2e-6
    assert rel(gx, x.grad) < 2e-5, rel(gx, x.grad)
    assert rel(dg, bn.weight.grad) < 2e-5 and rel(db, bn.bias.grad) < 2e-5
