"""GPU: each device stage of the fused GVA (ao_amd/csrc/gva_*.hip, through ao_amd/ptv2/gva.py's autograd
Functions) against the plain-torch statement of the same stage (tests/gva_torch_ref.py) on the GPU."""
import numpy as np
import pytest
import torch

from tests import synth
from tests.gva_torch_ref import TorchImpl

pytestmark = pytest.mark.gpu

SHAPES = [(48, 6, 16, 3000), (96, 12, 16, 1500), (192, 24, 16, 700), (384, 48, 16, 300), (48, 6, 8, 1000),
          (512, 64, 16, 130)]


@pytest.fixture(params=["point", "flat"], autouse=True)
def kernel_family(request, monkeypatch):
    """Every stage test runs twice: with the MFMA point kernels where they are the default, and with the flat
    one-lane-per-slot kernels (AO_AMD_BWD_STAGED=1) that remain the path for shapes the point kernels do not
    cover (K > 16, channel / group counts outside the instantiated set)."""
    if request.param == "flat":
        monkeypatch.setenv("AO_AMD_BWD_STAGED", "1")
    else:
        monkeypatch.delenv("AO_AMD_BWD_STAGED", raising=False)
    return request.param


def make(c, g, k, n, seed=0):
    from ao_amd import pointops

    torch.manual_seed(seed)
    xyz = torch.from_numpy(synth.room_cloud(n, seed=seed)).cuda()
    off = torch.tensor([n // 3, n], dtype=torch.int32).cuda()
    idx, _ = pointops.knn_query(k, xyz, off)
    idx = idx.clone()
    idx[2::7, k - 2:] = -1
    return xyz, idx


def close(a, b, rtol=1e-4, atol=1e-5, name=""):
    a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
    scale = max(np.abs(b).max(), 1e-6)
    err = np.abs(a - b).max()
    assert err <= atol * max(1.0, scale) + rtol * scale, (name, err, scale)


@pytest.mark.parametrize("c,g,k,n", SHAPES)
def test_pos_stats_and_logits(c, g, k, n):
    from ao_amd.ptv2.gva import _HipImpl

    xyz, idx = make(c, g, k, n)
    s1, s2 = _HipImpl.pos_stats(xyz, idx)
    r1, r2 = TorchImpl.pos_stats(xyz, idx)
    close(s1, r1, 1e-5, 1e-6, "s1")
    close(s2, r2, 1e-5, 1e-6, "s2")
    args = [torch.randn(n, g), torch.randn(n, g), torch.randn(c, 3) * 3, torch.randn(c) * 0.3,
            torch.randn(c, g) * 0.2, torch.randn(g)]
    args = [t.cuda().requires_grad_(True) for t in args]
    W1, T1, T2 = _HipImpl.logits(*args, xyz, idx)
    rW1, rT1, rT2 = TorchImpl.logits(*args, xyz, idx)
    close(W1, rW1, name="W1")
    close(T1, rT1, 1e-5, 1e-6, "T1")
    close(T2, rT2, 1e-5, 1e-6, "T2")


@pytest.mark.parametrize("c,g,k,n", SHAPES)
def test_aggregate_forward(c, g, k, n):
    from ao_amd.ptv2.gva import _HipImpl

    xyz, idx = make(c, g, k, n, seed=1)
    args = [torch.randn(n, k, g), torch.rand(g) + 0.5, torch.randn(g) * 0.3, torch.randn(g, g) * 0.5, torch.randn(g),
            torch.randn(n, c), torch.randn(c, 3) * 3, torch.randn(c) * 0.3]
    args = [t.cuda() for t in args]
    out_v, A, sw = _HipImpl.aggregate(*args, xyz, idx)
    r_out, r_A, r_sw = TorchImpl.aggregate(*args, xyz, idx)
    close(out_v, r_out, name="out_v")
    close(A, r_A, name="A")
    close(sw, r_sw, name="sw")


def grads_close(out, ref, ins, gouts, names, rel=2e-3):
    g1 = torch.autograd.grad(out, ins, gouts, allow_unused=True)
    g2 = torch.autograd.grad(ref, ins, gouts, allow_unused=True)
    for n_, a, b in zip(names, g1, g2):
        a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
        err = np.linalg.norm(a - b)
        if n_ == "bw2":  # softmax is shift invariant: the true gradient is exactly zero, both sides hold rounding noise
            assert np.abs(a).max() < 1e-3 and np.abs(b).max() < 1e-3
            continue
        assert err <= rel * np.linalg.norm(b) + 1e-5, (n_, err, np.linalg.norm(b))
        assert np.abs(a - b).max() <= 2e-2 * max(np.abs(b).max(), 1e-6) + 1e-5, (n_, np.abs(a - b).max(), np.abs(b).max())


@pytest.mark.parametrize("c,g,k,n", SHAPES)
def test_logits_backward(c, g, k, n):
    from ao_amd.ptv2.gva import _HipImpl

    xyz, idx = make(c, g, k, n, seed=2)
    names = ["kW", "qW", "a", "b", "M", "cW"]
    ins = [torch.randn(n, g), torch.randn(n, g), torch.randn(c, 3) * 3, torch.randn(c) * 0.3, torch.randn(c, g) * 0.2,
           torch.randn(g)]
    ins = [t.cuda().requires_grad_(True) for t in ins]
    out = _HipImpl.logits(*ins, xyz, idx)
    ref = TorchImpl.logits(*ins, xyz, idx)
    gouts = [torch.randn(n, k, g).cuda(), torch.randn(g).double().cuda() * 0.1, torch.randn(g).double().cuda() * 0.01]
    grads_close(out, ref, ins, gouts, names)


@pytest.mark.parametrize("c,g,k,n", SHAPES)
def test_aggregate_backward(c, g, k, n):
    from ao_amd.ptv2.gva import _HipImpl

    xyz, idx = make(c, g, k, n, seed=3)
    names = ["W1", "sc", "sh", "Ww2", "bw2", "v", "a", "b"]
    ins = [torch.randn(n, k, g), torch.rand(g) + 0.5, torch.randn(g) * 0.3, torch.randn(g, g) * 0.5, torch.randn(g),
           torch.randn(n, c), torch.randn(c, 3) * 3, torch.randn(c) * 0.3]
    ins = [t.cuda().requires_grad_(True) for t in ins]
    out = _HipImpl.aggregate(*ins, xyz, idx)
    ref = TorchImpl.aggregate(*ins, xyz, idx)
    gouts = [torch.randn(n, c).cuda(), torch.randn(n, g, c).cuda(), torch.randn(n, g).cuda()]
    grads_close(out, ref, ins, gouts, names)


def test_inverse_table_matches_host_statement():
    from ao_amd.ptv2.gva import inverse_table

    xyz, idx = make(48, 6, 16, 5000, seed=4)
    ptr, rows = inverse_table(idx)
    hptr, hrows = inverse_table(idx.cpu())
    assert torch.equal(ptr.cpu(), hptr) and torch.equal(rows.cpu(), hrows)


def _host_inverse(idx):
    from ao_amd.ptv2.gva import inverse_table

    return inverse_table(idx.cpu().clone())


def test_inverse_tables_of_a_scene_in_one_call():
    """gva.inverse_tables: tables of different shapes side by side in one native call (csrc/inverse.hip: counting sort in five
    launches) -- self tables, interpolation tables (k = 3, targets in a coarser level: most buckets empty), placeholders, a
    hub that many rows point at (buckets beyond the 16-lane path), single-row and single-column tables, more tables than one
    call takes, and the bench's size; every list bit-equal to the stable sort of the host statement."""
    from ao_amd.ptv2 import gva

    g = torch.Generator().manual_seed(12)
    tables = []
    for n, k, hi in [(5000, 16, 5000), (5000, 8, 5000), (20000, 3, 4800), (1, 16, 1), (777, 1, 777), (120000, 16, 120000),
                     (30000, 3, 7000), (257, 16, 257), (4097, 4, 4097)]:
        idx = torch.randint(0, hi, (n, k), generator=g, dtype=torch.int32)
        if n > 100:
            idx[3::7, k - 1] = -1            # placeholders (bucket 0: n / 7 members)
            idx[::5, 0] = min(hi, 42) - 1    # a hub: n / 5 rows point at one target
            idx[1::50, 0] = hi - 1           # the last target
        tables.append(idx.cuda())
    small = [torch.randint(-1, 50, (50, 2), generator=g, dtype=torch.int32).cuda() for _ in range(20)]  # > 16 jobs: two calls
    res = gva.inverse_tables(tables + small)
    assert len(res) == len(tables) + len(small)
    for idx, (ptr, rows) in zip(tables + small, res):
        hptr, hrows = _host_inverse(idx)
        assert torch.equal(ptr.cpu(), hptr), idx.shape
        assert torch.equal(rows.cpu(), hrows), idx.shape
        assert gva.inverse_table(idx)[0] is ptr  # cached on the table
    # the single-table entry point and a second call on fresh copies give the same bits (arrival order does not show)
    for idx, (ptr, rows) in zip(tables[:3], res[:3]):
        p2, r2 = gva.inverse_table(idx.clone())
        assert torch.equal(p2, ptr) and torch.equal(r2, rows)
    # out-of-table entries are treated as placeholders, not written through
    bad = tables[0].clone()
    bad[7, 3], bad[9, 1] = 5000, -7
    ptr, rows = gva.inverse_table(bad)
    fixed = bad.clone()
    fixed[7, 3], fixed[9, 1] = -1, -1
    hptr, hrows = _host_inverse(fixed)
    assert torch.equal(ptr.cpu(), hptr) and torch.equal(rows.cpu(), hrows)


@pytest.mark.parametrize("n,c,g", [(3000, 48, 6), (500, 192, 24), (129, 384, 48), (40, 520, 65), (33, 40, 5), (1, 96, 12)])
def test_grouped_projection_kernels(n, c, g):
    """out = out_v + A Wp2_g^T + bp2 * sw and its backward (g_A, g_sw) against torch, incl. shapes whose g * c / 4 does not
    divide into workgroups (generic backward kernel) and C > 256 (column blocks, co-resident grid)."""
    from ao_amd import _lib
    import ao_amd.ptv2.gva  # noqa: F401

    L = _lib.lib()
    torch.manual_seed(n + c)
    I = c // g
    A = torch.randn(n, g, c, device="cuda")
    Wp2 = torch.randn(c, c, device="cuda") / c ** 0.5
    bp2 = torch.randn(c, device="cuda")
    sw = torch.rand(n, g, device="cuda")
    out_v = torch.randn(n, c, device="cuda")
    out = torch.empty(n, c, device="cuda")
    _lib.check(L.gva_peb_forward_hip_launcher(n, c, g, A.data_ptr(), Wp2.data_ptr(), bp2.data_ptr(), sw.data_ptr(), out_v.data_ptr(),
                                              out.data_ptr(), _lib.stream_ptr()), "peb fwd")
    Wg = Wp2.view(g, I, c).double()
    ref = out_v.double() + torch.einsum("ngc,gic->ngi", A.double(), Wg).reshape(n, c) + bp2.double() * sw.double().repeat_interleave(I, 1)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    go = torch.randn(n, c, device="cuda")
    gA = torch.empty(n, g, c, device="cuda")
    gsw = torch.empty(n, g, device="cuda")
    _lib.check(L.gva_peb_backward_hip_launcher(n, c, g, go.data_ptr(), Wp2.data_ptr(), bp2.data_ptr(), gA.data_ptr(), gsw.data_ptr(),
                                               _lib.stream_ptr()), "peb bwd")
    gA_ref = torch.einsum("ngi,gic->ngc", go.double().view(n, g, I), Wg)
    gsw_ref = (go.double() * bp2.double()).view(n, g, I).sum(-1)
    np.testing.assert_allclose(gA.cpu().numpy(), gA_ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gsw.cpu().numpy(), gsw_ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
