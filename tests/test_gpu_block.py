"""GPU: the fp32 MFMA row GEMM (csrc/gemm.hip) against float64 matmul, and the native Block runtime
(csrc/block.hip: one call per direction) against the same Block evaluated op by op in python
(AO_AMD_BLOCK=python: rocBLAS linears + the staged HIP kernels), forward, every gradient, running statistics."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("m,n,k", [(120000, 48, 48), (18905, 96, 96), (4501, 192, 192), (1074, 384, 384), (777, 512, 512),
                                   (1000, 96, 48), (1000, 48, 96), (63, 64, 32), (65, 16, 4), (1, 48, 48), (3000, 384, 192)])
@pytest.mark.parametrize("kmajor", [False, True])
@pytest.mark.parametrize("form", ["default", "lds", "direct"])
def test_rows_gemm(m, n, k, kmajor, form, monkeypatch):
    """`form`: the launcher's own choice, the LDS-staged kernel everywhere, the direct kernel wherever it is instantiated
    (AO_AMD_GEMM is read per call)."""
    from ao_amd.ptv2.block import rows_gemm

    if form != "default":
        monkeypatch.setenv("AO_AMD_GEMM", form)
    torch.manual_seed(m + n + k)
    x = torch.randn(m, k, device="cuda")
    w = torch.randn((k, n) if kmajor else (n, k), device="cuda") / k ** 0.5
    b = torch.randn(n, device="cuda")
    ref = x.double() @ (w.double() if kmajor else w.double().t())
    y = rows_gemm(x, w, w_kmajor=kmajor)
    assert rel(y, ref) < 2e-6
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().cpu().numpy(), rtol=0, atol=2e-5)
    yb = rows_gemm(x, w, bias=b, w_kmajor=kmajor)
    np.testing.assert_allclose(yb.cpu().numpy(), (ref + b.double()).float().cpu().numpy(), rtol=0, atol=2e-5)
    acc = torch.randn(m, n, device="cuda")
    expect = (acc.double() + ref).float()
    rows_gemm(x, w, w_kmajor=kmajor, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), expect.cpu().numpy(), rtol=0, atol=2e-5)


def _ptr_array(tensors):
    import ctypes

    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


@pytest.mark.parametrize("m,c", [(120000, 48), (7500, 192), (1900, 384), (333, 512)])
def test_rows_gemm_multi(m, c):
    """q/k/v form (three independent outputs) and the summed form (g_f1 = sum_i gY_i W_i) of one launch."""
    from ao_amd import _lib
    import ao_amd.ptv2.block  # noqa: F401

    torch.manual_seed(c)
    L = _lib.lib()
    x = torch.randn(m, c, device="cuda")
    ws = [torch.randn(c, c, device="cuda") / c ** 0.5 for _ in range(3)]
    bs = [torch.randn(c, device="cuda") for _ in range(3)]
    ys = [torch.empty(m, c, device="cuda") for _ in range(3)]
    rc = L.rows_gemm_multi_hip_launcher(m, c, c, 3, 0, _ptr_array([x, x, x]), _ptr_array(ws), 0, _ptr_array(bs), _ptr_array(ys),
                                        0, _lib.stream_ptr())
    _lib.check(rc, "rows_gemm_multi_hip_launcher")
    for w, b, y in zip(ws, bs, ys):
        ref = (x.double() @ w.double().t() + b.double()).float()
        np.testing.assert_allclose(y.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=3e-5)
    gys = [torch.randn(m, c, device="cuda") for _ in range(3)]
    out = torch.empty(m, c, device="cuda")
    rc = L.rows_gemm_multi_hip_launcher(m, c, c, 3, 1, _ptr_array(gys), _ptr_array(ws), 1, None, _ptr_array([out, None, None]), 0,
                                        _lib.stream_ptr())
    _lib.check(rc, "rows_gemm_multi_hip_launcher")
    ref = sum(g.double() @ w.double() for g, w in zip(gys, ws)).float()
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=6e-5)


@pytest.mark.parametrize("n,cout,cin,count", [(120000, 48, 48, 3), (7500, 24, 192, 2), (1900, 384, 384, 3), (4097, 6, 48, 2)])
def test_linear_wgrad_multi(n, cout, cin, count):
    from ao_amd import _lib

    torch.manual_seed(n)
    L = _lib.lib()
    gys = [torch.randn(n, cout, device="cuda") for _ in range(count)]
    xs = [torch.randn(n, cin, device="cuda") for _ in range(count)]
    dws = [torch.empty(cout, cin, device="cuda") for _ in range(count)]
    dbs = [torch.empty(cout, device="cuda") if i != 1 else None for i in range(count)]
    ws = _lib.workspace(L.dense_workspace_bytes(n, count * cout, cin), xs[0].device)
    rc = L.linear_wgrad_multi_hip_launcher(n, cout, cin, count, _ptr_array(gys), _ptr_array(xs), _ptr_array(dws), _ptr_array(dbs), None, None,
                                           ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "linear_wgrad_multi_hip_launcher")
    for gy, x, dw, db in zip(gys, xs, dws, dbs):
        ref = gy.double().t() @ x.double()
        assert rel(dw, ref) < 1e-5
        if db is not None:
            assert rel(db, gy.double().sum(0)) < 1e-5


@pytest.mark.parametrize("n,c", [(120000, 48), (30001, 96), (4501, 192), (1074, 384), (70, 512)])
def test_batchnorm_fused_into_gemm(n, c):
    """GEMM-epilogue statistics + bn_tiles_finalize against nn.BatchNorm1d on the GEMM output, and the normalise +
    ReLU applied on the operand load of the next GEMM / weight gradient / G-wide projection against the explicit
    sequence."""
    import ctypes

    from ao_amd import _lib
    import ao_amd.ptv2.block  # noqa: F401

    torch.manual_seed(n)
    L = _lib.lib()
    s = _lib.stream_ptr()
    x = torch.randn(n, c, device="cuda")
    w1 = torch.randn(c, c, device="cuda") / c ** 0.5
    b1 = torch.randn(c, device="cuda") * 3          # |mean| >> std in some columns: the variance must survive
    h = torch.empty(n, c, device="cuda")
    st = torch.empty(L.bn_tiles_floats(n, c), device="cuda")
    rc = L.rows_gemm_fused_hip_launcher(n, c, c, 1, 0, _ptr_array([x]), _ptr_array([w1]), 0, _ptr_array([b1]), _ptr_array([h]), 0,
                                        None, None, _ptr_array([st]), s)
    _lib.check(rc, "rows_gemm_fused_hip_launcher")
    bn = torch.nn.BatchNorm1d(c).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
    ref_rm, ref_rv = bn.running_mean.clone(), bn.running_var.clone()
    mean, rstd, sc, sh = (torch.empty(c, device="cuda") for _ in range(4))
    rm, rv = ref_rm.clone(), ref_rv.clone()
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    rc = L.bn_tiles_finalize_hip_launcher(n, c, st.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), sc.data_ptr(), sh.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                          nbt.data_ptr(), float(bn.eps), float(bn.momentum), s)
    _lib.check(rc, "bn_tiles_finalize_hip_launcher")
    f_ref = torch.relu(bn(h))                       # also updates bn.running_*
    hd = h.double()
    np.testing.assert_allclose(mean.cpu().numpy(), hd.mean(0).float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rstd.cpu().numpy(), (hd.var(0, unbiased=False) + bn.eps).rsqrt().float().cpu().numpy(), rtol=2e-5)
    np.testing.assert_allclose(rm.cpu().numpy(), bn.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), bn.running_var.cpu().numpy(), rtol=1e-4, atol=1e-6)
    assert int(nbt) == 1
    # consumers with the fused normalise + ReLU
    w2 = torch.randn(c, c, device="cuda") / c ** 0.5
    y = torch.empty(n, c, device="cuda")
    rc = L.rows_gemm_fused_hip_launcher(n, c, c, 1, 0, _ptr_array([h]), _ptr_array([w2]), 0, None, _ptr_array([y]), 0,
                                        sc.data_ptr(), sh.data_ptr(), None, s)
    _lib.check(rc, "rows_gemm_fused_hip_launcher")
    np.testing.assert_allclose(y.cpu().numpy(), (f_ref.detach() @ w2.t()).cpu().numpy(), rtol=1e-4, atol=2e-4)
    gy = torch.randn(n, c, device="cuda")
    dW = torch.empty(c, c, device="cuda")
    ws = _lib.workspace(L.dense_workspace_bytes(n, c, c), x.device)
    rc = L.linear_wgrad_multi_hip_launcher(n, c, c, 1, _ptr_array([gy]), _ptr_array([h]), _ptr_array([dW]), None, _ptr_array([sc]),
                                           _ptr_array([sh]), ws.data_ptr(), ws.numel(), s)
    _lib.check(rc, "linear_wgrad_multi_hip_launcher")
    assert rel(dW, gy.double().t() @ f_ref.detach().double()) < 1e-5
    g = max(c // 8, 4)
    ww = torch.randn(g, c, device="cuda") / c ** 0.5
    kw = torch.empty(n, g, device="cuda")
    rc = L.skinny_linear_forward_xf_hip_launcher(n, c, g, h.data_ptr(), ww.data_ptr(), sc.data_ptr(), sh.data_ptr(), kw.data_ptr(), s)
    _lib.check(rc, "skinny_linear_forward_xf_hip_launcher")
    np.testing.assert_allclose(kw.cpu().numpy(), (f_ref.detach() @ ww.t()).cpu().numpy(), rtol=1e-4, atol=2e-4)


def _block_pair(c, g, drop, seed):
    from ao_amd.ptv2.model import Block

    torch.manual_seed(seed)
    a = Block(c, g, drop_path_rate=drop).cuda()
    with torch.no_grad():
        for name, p in a.named_parameters():
            if p.dim() == 1 and "norm" in name and name.endswith("weight"):
                p.uniform_(0.6, 1.4)
            elif p.dim() == 1:
                p.normal_(0, 0.2)
        for name, buf in a.named_buffers():
            if name.endswith("running_mean"):
                buf.normal_(0, 0.3)
            elif name.endswith("running_var"):
                buf.uniform_(0.5, 2.0)
    return a, copy.deepcopy(a)


@pytest.mark.parametrize("n,c,g,k", [(6000, 48, 6, 16), (3000, 96, 12, 16), (1500, 192, 24, 16), (700, 384, 48, 16),
                                     (2000, 48, 6, 8), (300, 512, 64, 16)])
@pytest.mark.parametrize("training", [True, False])
def test_native_block_matches_python_block(monkeypatch, n, c, g, k, training):
    from ao_amd import pointops, synth
    from ao_amd.ptv2 import block as native

    cloud = synth.room_cloud(n, seed=n + c)
    coord = torch.from_numpy(cloud).cuda()
    offset = torch.tensor([n], dtype=torch.int32, device="cuda")
    idx, _ = pointops.knn_query(k, coord, offset)
    torch.manual_seed(7)
    x0 = torch.randn(n, c, device="cuda").relu_()
    go = torch.randn(n, c, device="cuda")
    blk_n, blk_p = _block_pair(c, g, 0.0, seed=3)
    outs = {}
    for tag, blk in (("native", blk_n), ("python", blk_p)):
        monkeypatch.setenv("AO_AMD_BLOCK", tag)
        blk.train(training)
        x = x0.clone().requires_grad_(True)
        if tag == "native":
            assert native.supported(blk, x, idx)
        y = blk([coord, x, offset], idx)[1]
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), go)
        outs[tag] = (y.detach(), grads, {k_: v.clone() for k_, v in blk.state_dict().items()})
    y_n, g_n, sd_n = outs["native"]
    y_p, g_p, sd_p = outs["python"]
    np.testing.assert_allclose(y_n.cpu().numpy(), y_p.cpu().numpy(), rtol=1e-4, atol=1e-4)
    names = ["x"] + [nm for nm, _ in blk_n.named_parameters()]
    for nm, a, b in zip(names, g_n, g_p):
        # biases whose effect a training-mode BatchNorm (or the softmax's shift invariance) removes have an exactly
        # zero gradient that both sides compute as O(1e-3) summation noise over n*k rows: absolute floor for those
        zero_grad = training and nm in ("attn.linear_q.0.bias", "attn.linear_k.0.bias", "attn.linear_v.bias",
                                        "attn.linear_p_bias.0.bias", "attn.linear_p_bias.3.bias",
                                        "attn.weight_encoding.0.bias", "attn.weight_encoding.3.bias")
        floor = 5e-3 if zero_grad or nm == "attn.weight_encoding.3.bias" else 1e-4
        # 1e-2: both sides are fp32 HIP paths with different summation orders; a 1e-6 difference in a pre-activation
        # flips ReLU masks and moves whole terms of the BatchNorm parameter gradients
        assert rel(a, b) < 1e-2 or float((a - b).abs().max()) < floor, (nm, rel(a, b), float((a - b).abs().max()))
    for key in sd_p:
        np.testing.assert_allclose(sd_n[key].cpu().numpy(), sd_p[key].cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=key)


def test_native_block_droppath_rowscale(monkeypatch):
    """With DropPath the dropped points pass the identity through unchanged (ReLU of a ReLU output); same draw of
    the per-point mask -> same output and input gradient as the op-by-op block."""
    from ao_amd import pointops, synth

    n, c, g, k = 5000, 48, 6, 16
    coord = torch.from_numpy(synth.room_cloud(n, seed=5)).cuda()
    offset = torch.tensor([n], dtype=torch.int32, device="cuda")
    idx, _ = pointops.knn_query(k, coord, offset)
    blk_n, blk_p = _block_pair(c, g, 0.5, seed=11)
    x0 = torch.randn(n, c, device="cuda").relu_()
    res = {}
    for tag, blk in (("native", blk_n), ("python", blk_p)):
        monkeypatch.setenv("AO_AMD_BLOCK", tag)
        blk.train()
        x = x0.clone().requires_grad_(True)
        torch.manual_seed(99)
        y = blk([coord, x, offset], idx)[1]
        (gx,) = torch.autograd.grad((y * y).sum(), [x])
        res[tag] = (y.detach(), gx)
    y, gx = res["native"]
    same = (y == x0).all(dim=1)
    assert 0.4 < float(same.float().mean()) < 0.6
    np.testing.assert_allclose(y.cpu().numpy(), res["python"][0].cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert rel(gx, res["python"][1]) < 5e-3


@pytest.mark.parametrize("c,g", [(48, 6), (96, 12), (192, 24), (384, 48)])
def test_native_block_with_short_clouds(monkeypatch, c, g):
    """A batch that contains clouds with fewer than K points: their neighbour tables carry -1 placeholders, which
    every fused stage has to mask exactly as the reference does (softmax over all K slots, mask afterwards)."""
    from ao_amd import pointops, synth
    from ao_amd.ptv2 import block as native

    k = 16
    sizes = [700, 9, 1500, 3, 16, 800]
    coord = torch.from_numpy(np.concatenate([synth.room_cloud(max(n, 64), seed=i)[:n] for i, n in enumerate(sizes)])).cuda()
    offset = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device="cuda")
    n = int(offset[-1])
    idx, _ = pointops.knn_query(k, coord, offset)
    assert int((idx < 0).sum()) > 0
    torch.manual_seed(1)
    x0 = torch.randn(n, c, device="cuda").relu_()
    go = torch.randn(n, c, device="cuda")
    blk_n, blk_p = _block_pair(c, g, 0.0, seed=9)
    res = {}
    for tag, blk in (("native", blk_n), ("python", blk_p)):
        monkeypatch.setenv("AO_AMD_BLOCK", tag)
        monkeypatch.setenv("AO_AMD_GVA", "fused" if tag == "native" else "unfused")  # python side: the literal op sequence
        blk.train()
        x = x0.clone().requires_grad_(True)
        if tag == "native":
            assert native.supported(blk, x, idx)
        y = blk([coord, x, offset], idx)[1]
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), go)
        res[tag] = (y.detach(), grads)
    np.testing.assert_allclose(res["native"][0].cpu().numpy(), res["python"][0].cpu().numpy(), rtol=1e-4, atol=1e-4)
    names = ["x"] + [nm for nm, _ in blk_n.named_parameters()]
    zero_grad = ("attn.linear_q.0.bias", "attn.linear_k.0.bias", "attn.linear_v.bias", "attn.linear_p_bias.0.bias",
                 "attn.linear_p_bias.3.bias", "attn.weight_encoding.0.bias", "attn.weight_encoding.3.bias")
    for nm, a, b in zip(names, res["native"][1], res["python"][1]):
        # biases whose true gradient is exactly zero: the literal op sequence returns O(1e-2) summation noise over
        # the n*k rows, the fused path (closed-form BN fold) returns zero or its own noise
        floor = 5e-2 if nm in zero_grad else 1e-4
        assert rel(a, b) < 5e-3 or float((a - b).abs().max()) < floor, (nm, rel(a, b), float((a - b).abs().max()))


@pytest.mark.parametrize("n,c,g", [(20000, 48, 6), (3000, 192, 24)])
def test_native_block_is_bitwise_reproducible(monkeypatch, n, c, g):
    """No float atomics on the Block path: scatter-adds are inverse-table gathers, every reduction is per-block
    partials summed in a fixed order -- two runs give bit-identical outputs and gradients."""
    from ao_amd import pointops, synth

    monkeypatch.setenv("AO_AMD_BLOCK", "native")
    coord = torch.from_numpy(synth.room_cloud(n, seed=3)).cuda()
    offset = torch.tensor([n], dtype=torch.int32, device="cuda")
    idx, _ = pointops.knn_query(16, coord, offset)
    blk, _ = _block_pair(c, g, 0.0, seed=2)
    blk.train()
    torch.manual_seed(5)
    x0 = torch.randn(n, c, device="cuda").relu_()
    go = torch.randn(n, c, device="cuda")
    runs = []
    for _ in range(2):
        x = x0.clone().requires_grad_(True)
        y = blk([coord, x, offset], idx)[1]
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), go)
        runs.append([y.detach()] + [t.clone() for t in grads])
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_native_block_under_autocast(monkeypatch):
    """The reference trainer wraps the step in torch.cuda.amp.autocast (reference pointcept/engines/train.py:178): the Block
    runtime is still taken there; its Linear products then run on the bf16 matrix cores (operands rounded to bf16, fp32
    accumulation -- autocast's own arithmetic for nn.Linear) while the activations stay fp32.  AO_AMD_AUTOCAST_MATMUL=fp32
    keeps exact fp32 products: the same bits as without autocast."""
    from ao_amd import pointops, synth
    from ao_amd.ptv2 import block as native

    monkeypatch.setenv("AO_AMD_BLOCK", "native")
    n, c, g = 4000, 96, 12
    coord = torch.from_numpy(synth.room_cloud(n, seed=8)).cuda()
    offset = torch.tensor([n], dtype=torch.int32, device="cuda")
    idx, _ = pointops.knn_query(16, coord, offset)
    blk, _ = _block_pair(c, g, 0.0, seed=4)
    blk.train()
    torch.manual_seed(6)
    x0 = torch.randn(n, c, device="cuda").relu_()
    go = torch.randn(n, c, device="cuda")
    runs = {}
    for tag, amp, matmul in (("fp32", False, "bf16"), ("amp_bf16", True, "bf16"), ("amp_fp32", True, "fp32")):
        monkeypatch.setenv("AO_AMD_AUTOCAST_MATMUL", matmul)
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            assert native.supported(blk, x, idx)
            y = blk([coord, x, offset], idx)[1]
        assert y.dtype == torch.float32
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), go)
        runs[tag] = [y.detach()] + [t.clone() for t in grads]
    for a, b in zip(runs["fp32"], runs["amp_fp32"]):
        assert torch.equal(a, b)
    y32, yb = runs["fp32"][0], runs["amp_bf16"][0]
    assert not torch.equal(y32, yb) and rel(yb, y32) < 2e-2  # bf16 products ran; one Block stays within 2e-2 relative L2
    assert rel(runs["amp_bf16"][1], runs["fp32"][1]) < 0.1   # input gradient


def test_plan_follows_rehomed_parameters_and_buffers():
    """the cached native plan holds raw device pointers of all 30 parameters and 21 BatchNorm buffers; it
    must be rebuilt when ANY of them moves -- FlatAdamW re-homing every trainable `p.data` while fc1 is frozen,
    load_state_dict(assign=True), a replaced running buffer."""
    import ao_amd.ptv2 as ptv2
    from ao_amd import pointops
    from ao_amd.ptv2.optim import FlatAdamW

    torch.manual_seed(3)
    blk = ptv2.Block(48, 6).cuda().eval()
    n = 1500
    coord = torch.rand(n, 3, device="cuda")
    off = torch.tensor([n], dtype=torch.int32, device="cuda")
    idx = pointops.knn_query(16, coord, off)[0]
    feat = torch.randn(n, 48, device="cuda")
    with torch.no_grad():
        y0 = blk([coord, feat, off], idx)[1].clone()
        # 1) fc1 frozen, everything else re-homed into a flat buffer by the optimizer
        blk.fc1.weight.requires_grad_(False)
        opt = FlatAdamW(blk.parameters(), lr=0.1)
        assert torch.equal(blk([coord, feat, off], idx)[1], y0)
        opt.flat_param.mul_(1.5)  # the parameters now live here: the output must follow
        y1 = blk([coord, feat, off], idx)[1].clone()
        assert not torch.allclose(y1, y0)
        twin = ptv2.Block(48, 6).cuda().eval()
        twin.load_state_dict(copy.deepcopy(blk.state_dict()))
        assert torch.allclose(twin([coord, feat, off], idx)[1], y1, atol=1e-6)
        # 2) a replaced BatchNorm buffer
        blk.norm2.norm.running_var = blk.norm2.norm.running_var * 4.0
        y2 = blk([coord, feat, off], idx)[1].clone()
        twin.load_state_dict(copy.deepcopy(blk.state_dict()))
        assert not torch.allclose(y2, y1) and torch.allclose(twin([coord, feat, off], idx)[1], y2, atol=1e-6)
        # 3) load_state_dict(assign=True) swaps the Parameter objects themselves
        sd = {k: v.clone() * (0.5 if k == "attn.linear_v.weight" else 1.0) for k, v in blk.state_dict().items()}
        blk.load_state_dict(sd, assign=True)
        y3 = blk([coord, feat, off], idx)[1].clone()
        twin.load_state_dict(copy.deepcopy(blk.state_dict()))
        assert not torch.allclose(y3, y2) and torch.allclose(twin([coord, feat, off], idx)[1], y3, atol=1e-6)


@pytest.mark.parametrize("n,c,g", [(3000, 96, 12), (4501, 192, 24), (1074, 384, 48), (20011, 48, 6), (51, 48, 6)])
def test_fused_logits_backward_equals_the_staged_kernels(monkeypatch, n, c, g):
    """gva_bwd_logits.hip (round 3): rows + parameter gradients of the logits stage in one pipelined MFMA launch, against the
    three staged kernels of gva_bwd.hip (AO_AMD_LOGITS_BWD=staged) inside the same native Block.  The row gradient gWt is
    the same fused multiply-add in both, so everything downstream of it -- the input gradient, the q / k projections -- is
    bit-identical; the parameter gradients of the stage (grad M -> linear_p_bias / weight_encoding, BN_p, grad cW) differ in
    summation order only.  Includes clouds shorter than K (masked -1 slots)."""
    from ao_amd import pointops, synth

    k = 16
    sizes = [n - 30, 9, 21]
    coord = torch.from_numpy(np.concatenate([synth.room_cloud(max(m, 64), seed=40 + i)[:m] for i, m in enumerate(sizes)])).cuda()
    offset = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device="cuda")
    idx, _ = pointops.knn_query(k, coord, offset)
    torch.manual_seed(3)
    x0 = torch.randn(n, c, device="cuda").relu_()
    go = torch.randn(n, c, device="cuda")
    blk, _ = _block_pair(c, g, 0.0, seed=21)
    blk.train()
    res = {}
    for mode in ("fused", "staged"):
        monkeypatch.setenv("AO_AMD_LOGITS_BWD", mode)
        x = x0.clone().requires_grad_(True)
        y = blk([coord, x, offset], idx)[1]
        res[mode] = torch.autograd.grad(y, [x] + list(blk.parameters()), go)
    names = ["x"] + [nm for nm, _ in blk.named_parameters()]
    assert torch.equal(res["fused"][0], res["staged"][0])  # the chain through gWt is the same arithmetic
    # biases in front of a training-mode BatchNorm / the softmax have an exactly-zero true gradient: both forms return the
    # O(1e-4) rounding noise of a sum of ~50 k cancelling terms (grad cW), in different orders
    zero_grad = ("attn.linear_q.0.bias", "attn.linear_k.0.bias", "attn.linear_v.bias", "attn.linear_p_bias.0.bias",
                 "attn.linear_p_bias.3.bias", "attn.weight_encoding.0.bias", "attn.weight_encoding.3.bias")
    for nm, a, b in zip(names, res["fused"], res["staged"]):
        floor = 1e-3 if nm in zero_grad else 2e-6
        assert rel(a, b) < 2e-5 or float((a - b).abs().max()) < floor, (nm, rel(a, b), float((a - b).abs().max()))


@pytest.mark.parametrize("n,c,g", [(4501, 192, 24), (1074, 384, 48), (9000, 96, 12), (5000, 48, 6)])
def test_bn_backward_finalize_in_the_apply_kernel_equals_the_three_launch_form(monkeypatch, n, c, g):
    """dense.hip (round 3): at the deep levels the BatchNorm backward's record sum runs in the prologue of the apply kernel
    (bn_bwd_finapply_kernel) instead of a finalize launch of its own -- all four BatchNorm backwards of a Block (norm3 with the
    residual tail, norm2 and norm1 from the GEMM-epilogue records, the linear_q / linear_k pair).  Against
    AO_AMD_BN_FINAPPLY=0 (reduce -> finalize -> apply): same arithmetic per element, the column sums differ in summation
    order only."""
    from ao_amd import pointops, synth

    k = 16
    coord = torch.from_numpy(synth.room_cloud(n, seed=n)).cuda()
    offset = torch.tensor([n], dtype=torch.int32, device="cuda")
    idx, _ = pointops.knn_query(k, coord, offset)
    torch.manual_seed(5)
    x0 = torch.randn(n, c, device="cuda").relu_()
    go = torch.randn(n, c, device="cuda")
    blk, _ = _block_pair(c, g, 0.3, seed=31)
    blk.train()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("AO_AMD_BN_FINAPPLY", mode)
        torch.manual_seed(77)  # the DropPath draw
        x = x0.clone().requires_grad_(True)
        y = blk([coord, x, offset], idx)[1]
        res[mode] = torch.autograd.grad(y, [x] + list(blk.parameters()), go)
    names = ["x"] + [nm for nm, _ in blk.named_parameters()]
    zero_grad = ("attn.linear_q.0.bias", "attn.linear_k.0.bias", "attn.linear_v.bias", "attn.linear_p_bias.0.bias",
                 "attn.linear_p_bias.3.bias", "attn.weight_encoding.0.bias", "attn.weight_encoding.3.bias")
    for nm, a, b in zip(names, res["1"], res["0"]):
        floor = 5e-3 if nm in zero_grad else 2e-6
        assert rel(a, b) < 5e-5 or float((a - b).abs().max()) < floor, (nm, rel(a, b), float((a - b).abs().max()))


@pytest.mark.parametrize("n", [20011, 6000, 129, 51])
@pytest.mark.parametrize("training", [True, False])
def test_fused_attention_forward_equals_the_three_staged_launches(monkeypatch, n, training):
    """gva_fwd_point.hip (round 5): softmax + aggregation + grouped projection of the full-resolution attention in one launch
    (one point per wavefront, the column statistics for norm2 in its epilogue), against the three staged launches
    (AO_AMD_FWD_STAGED=1: softmax_rows, aggregate_tile, peb_fwd_mfma) inside the same native Block: the output, every gradient
    (w, sw and A are read by the backward) and the running statistics of the BatchNorm behind the attention.  The fused
    softmax uses the hardware exp2 / reciprocal (1e-6 relative) and sums in another order.  Includes clouds shorter than K
    (masked -1 slots) and row counts that are not multiples of the 64-row blocks the kernel walks."""
    from ao_amd import pointops, synth

    k, c, g = 16, 48, 6
    sizes = [n - 30, 9, 21]
    coord = torch.from_numpy(np.concatenate([synth.room_cloud(max(m, 64), seed=70 + i)[:m] for i, m in enumerate(sizes)])).cuda()
    offset = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device="cuda")
    idx, _ = pointops.knn_query(k, coord, offset)
    torch.manual_seed(9)
    x0 = torch.randn(n, c, device="cuda").relu_()
    go = torch.randn(n, c, device="cuda")
    blk0, _ = _block_pair(c, g, 0.0, seed=41)
    res, stats = {}, {}
    for mode in ("fused", "staged"):
        if mode == "staged":
            monkeypatch.setenv("AO_AMD_FWD_STAGED", "1")
        else:
            monkeypatch.delenv("AO_AMD_FWD_STAGED", raising=False)
        blk = copy.deepcopy(blk0)
        blk.train(training)
        x = x0.clone().requires_grad_(True)
        y = blk([coord, x, offset], idx)[1]
        res[mode] = [y.detach()] + list(torch.autograd.grad(y, [x] + list(blk.parameters()), go))
        stats[mode] = {nm: b.clone() for nm, b in blk.named_buffers()}
    names = ["y", "x"] + [nm for nm, _ in blk0.named_parameters()]
    zero_grad = ("attn.linear_q.0.bias", "attn.linear_k.0.bias", "attn.linear_v.bias", "attn.linear_p_bias.0.bias",
                 "attn.linear_p_bias.3.bias", "attn.weight_encoding.0.bias", "attn.weight_encoding.3.bias")
    for nm, a, b in zip(names, res["fused"], res["staged"]):
        if nm == "y":
            assert rel(a, b) < 2e-6, (nm, rel(a, b))
            continue
        # a gradient is a discontinuous function of the forward: where a pre-activation of norm2's ReLU lies within the 1e-6 the
        # two forwards differ by, its mask flips, and the entries the flipped element reaches through the attention backward
        # -- its 16 neighbours' rows -- move (seen: two flips among 20 011 x 48 elements, 0.16 % of grad x).  So: all but 1 %
        # of the entries agree tightly, and the whole tensor to 2e-3.
        floor = 2e-3 if nm in zero_grad else 5e-6
        if nm == "x":  # (a parameter gradient sums over the rows: a flip moves every entry a little)
            tight = ((a - b).abs() <= 2e-5 * b.abs() + 1e-5 * float(b.abs().max())).float().mean()
            assert float(tight) >= 0.99, (nm, float(tight))
        assert rel(a, b) < 2e-3 or float((a - b).abs().max()) < floor, (nm, rel(a, b), float((a - b).abs().max()))
    for nm in stats["fused"]:
        a, b = stats["fused"][nm].double(), stats["staged"][nm].double()
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), nm
