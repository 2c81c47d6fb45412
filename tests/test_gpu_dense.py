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
