"""GPU: the bf16 matrix-core path (BASELINE.json configs[2], configs[4]: "bf16").  Under torch.autocast -- the reference
trainer's enable_amp, pointcept/engines/train_sam_pp2s.py:178-180 -- the nn.Linear products of PT-v2m2 run on
V_MFMA_F32_16X16X32_BF16: operands rounded to bf16 (round to nearest even), fp32 accumulation, fp32 in memory;
BatchNorm statistics, softmax, coordinates and kNN stay fp32 (autocast keeps them fp32 as well).

Tolerances (stated here, measured values printed):
  * a single product against the float64 product of the bf16-ROUNDED operands: 2e-6 relative (only the fp32 summation
    order differs) -- the kernel computes exactly what it says;
  * against the fp32 path (the thing bf16 rounding perturbs): one product 1e-2 relative L2 (2^-9 per operand, random
    signs over k = 48 .. 384 terms); whole-model logits 6e-2 relative L2 (measured 2-4e-2 on random-init logits of O(0.3)) and loss within 2e-2 after 15 blocks;
    the whole gradient keeps its direction (cosine > 0.9 with the fp32 gradient; measured 0.946), weight matrices 0.45
    relative L2 in the median (measured 0.31) -- tighter than torch.autocast's own deviation on the CPU oracle of the same
    network (0.87 / 0.49), which additionally stores the Linear outputs in bf16.  Random labels at random init make this a
    small, noisy gradient; the training test below checks what matters: the loss falls as in fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ptv2_ref as M

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture
def bf16_matmul():
    from ao_amd import _lib
    import ao_amd.ptv2.block  # noqa: F401

    prev = _lib.lib().ptv2_matmul_precision(1)
    yield
    _lib.lib().ptv2_matmul_precision(prev)


@pytest.mark.parametrize("m,n,k", [(120000, 48, 48), (4501, 192, 192), (1074, 384, 384), (1000, 96, 48), (333, 64, 32)])
@pytest.mark.parametrize("kmajor", [False, True])
def test_rows_gemm_bf16_operands(bf16_matmul, m, n, k, kmajor):
    from ao_amd.ptv2.block import rows_gemm

    torch.manual_seed(m + n)
    x = torch.randn(m, k, device="cuda")
    w = torch.randn((k, n) if kmajor else (n, k), device="cuda") / k ** 0.5
    b = torch.randn(n, device="cuda")
    y = rows_gemm(x, w, b, w_kmajor=kmajor)
    xr, wr = x.bfloat16().double(), w.bfloat16().double()
    exact = xr @ (wr if kmajor else wr.t()) + b.double()
    assert rel(y, exact) < 2e-6  # the product of the rounded operands, fp32 accumulation
    full = x.double() @ (w.double() if kmajor else w.double().t()) + b.double()
    r = rel(y, full)
    assert 1e-4 < r < 1e-2, r  # it IS a bf16 product (not fp32), within bf16's budget


@pytest.mark.parametrize("n,cout,cin", [(120000, 48, 48), (4501, 192, 192), (5000, 13, 48), (3000, 48, 6)])
def test_linear_wgrad_bf16_operands(bf16_matmul, n, cout, cin):
    from ao_amd import _lib

    torch.manual_seed(n)
    gy = torch.randn(n, cout, device="cuda")
    x = torch.randn(n, cin, device="cuda")
    dW = torch.empty(cout, cin, device="cuda")
    db = torch.empty(cout, device="cuda")
    L = _lib.lib()
    ws = _lib.workspace(L.dense_workspace_bytes(n, cout, cin), gy.device)
    rc = L.linear_wgrad_hip_launcher(n, cout, cin, gy.data_ptr(), x.data_ptr(), dW.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                     ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "linear_wgrad_hip_launcher")
    exact = gy.bfloat16().double().t() @ x.bfloat16().double()
    assert rel(dW, exact) < 5e-6
    assert rel(db, gy.double().sum(0)) < 1e-5  # the bias gradient is summed from the fp32 values
    assert 1e-4 < rel(dW, gy.double().t() @ x.double()) < 1e-2


@pytest.mark.parametrize("tag,points", [("s3dis", 20000), ("scannet", 8000), ("s3dis_bench", 120000), ("scannet_bench", 100000)])
def test_model_under_autocast_tracks_the_fp32_path(tag, points):
    # (*_bench: the shapes profiles/*_bench_bf16.json / *_bench_scannet_bf16.json time -- 1 x 120 000 and 2 x 100 000 points)
    bench = tag.endswith("_bench")
    tag = tag.split("_")[0]
    import ao_amd.ptv2 as ptv2
    from ao_amd import synth
    from ao_amd.ptv2 import native_model

    cfg = dict(M.S3DIS_CFG if tag == "s3dis" else M.SCANNET_CFG, drop_path_rate=0.0)
    seeds = [0] if (bench and tag == "s3dis") else ([0, 1] if bench else [2, 3])
    b = synth.scene_batch(seeds, point_max=points, in_channels=cfg["in_channels"], num_classes=cfg["num_classes"], room=1 if bench else 0)
    data = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    # labels that are a function of the input (height bands): a gradient with signal in it, not the noise of random labels
    data["segment"] = (data["coord"][:, 2] * 4).long().clamp(0, cfg["num_classes"] - 1)
    res = {}
    for mode in ("fp32", "bf16"):
        model = ptv2.PointTransformerV2(**cfg).cuda().train()
        model.load_state_dict(M.init_state(cfg, seed=29), strict=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "bf16"):
            assert native_model.matmul_bf16() == (mode == "bf16")
            logits = model(data)
            loss = F.cross_entropy(logits.float(), data["segment"], ignore_index=-1)
        assert logits.dtype == torch.float32  # activations stay fp32
        grads = torch.autograd.grad(loss, list(model.parameters()))
        res[mode] = (logits.detach(), float(loss.detach()), grads)
    (lf, loss_f, gf), (lb, loss_b, gb) = res["fp32"], res["bf16"]
    r = rel(lb, lf)
    names = [n for n, _ in model.named_parameters()]
    mats = [(rel(a, b), n) for n, a, b in zip(names, gb, gf) if b.dim() == 2 and float(b.norm()) > 1e-6]
    worst = max(mats)
    median = sorted(v for v, _ in mats)[len(mats) // 2]
    fa, fb = torch.cat([g.reshape(-1) for g in gb]).double(), torch.cat([g.reshape(-1) for g in gf]).double()
    cos = float(fa @ fb / (fa.norm() * fb.norm()))
    print("%s: logits rel L2 %.2e, loss %.5f vs %.5f, gradient cosine %.5f, weight-matrix gradients rel L2 median %.2e worst %.2e (%s)"
          % (tag, r, loss_b, loss_f, cos, median, *worst))
    # differs from fp32 (bf16 products really ran) and stays within what this build shows (round 5, both sizes of each cfg):
    #   S3DIS cfg   logits 3.9-4.2e-2, gradient cosine 0.969-0.977, weight-matrix gradients median 0.22-0.25 / worst 0.32-0.38
    #   ScanNet cfg logits 5.9-6.8e-2, gradient cosine 0.928-0.938, median 0.35-0.39 / worst 0.50-0.52 (its deepest levels hold
    #               a handful of points)
    # bounds: S3DIS cfg twice the observed deviation for the logits and 1 - cosine, 1.3 x for the per-matrix figures; ScanNet cfg
    # 1.2-1.5 x (its observed figures leave no room for a factor of two below "unrelated") (the old
    # `worst < 0.65` for both would have passed a badly broken layer of the S3DIS cfg; what pins a product is the operand-level
    # test above, 2e-6).  Context: torch.autocast(bfloat16) itself on the CPU oracle of this network (6 000 points, same seed)
    # deviates from its fp32 run by MORE -- logits 7.7e-2, cosine 0.87, matrices 0.49 median / 0.64 worst -- because it also
    # stores every Linear output in bf16.
    lim = dict(s3dis=dict(r=8.4e-2, cos=0.938, median=0.33, worst=0.50), scannet=dict(r=1.0e-1, cos=0.90, median=0.45, worst=0.65))[tag]
    assert 1e-5 < r < lim["r"], r
    assert abs(loss_b - loss_f) < 2e-2
    assert cos > lim["cos"] and median < lim["median"] and worst[0] < lim["worst"], (cos, median, worst)


def test_training_under_autocast_learns():
    """Ten optimizer steps under autocast on a small scene: the loss falls as it does in fp32."""
    import ao_amd.ptv2 as ptv2
    from ao_amd import synth
    from ao_amd.ptv2.optim import FlatAdamW

    b = synth.scene_batch([1], point_max=8000)
    data = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    data["segment"] = (data["coord"][:, 2] * 3).long().clamp(0, 12)  # learnable labels: a function of height
    curves = {}
    for mode in ("fp32", "bf16"):
        torch.manual_seed(0)
        seg = ptv2.DefaultSegmentor(dict(ptv2.S3DIS_BACKBONE, drop_path_rate=0.0)).cuda().train()
        seg.backbone.load_state_dict(M.init_state(dict(M.S3DIS_CFG), seed=3), strict=True)
        opt = FlatAdamW(seg.parameters(), lr=0.006, weight_decay=0.05)
        losses = []
        for _ in range(10):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "bf16"):
                loss = seg(data)["loss"]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        curves[mode] = losses
    print("fp32", ["%.3f" % v for v in curves["fp32"]], "\nbf16", ["%.3f" % v for v in curves["bf16"]])
    assert curves["bf16"][-1] < 0.7 * curves["bf16"][0]
    assert abs(curves["bf16"][-1] - curves["fp32"][-1]) < 0.15 * curves["fp32"][0]
