"""REAL's segmentor + logit basket on the GPU (BASELINE.json configs[3]) against the fixture captured from the
reference's DefaultSegmentorSAM_Image and trainer statement (tests/golden/segmentor_sam.npz;
pointcept/models/default.py:15-76, pointcept/engines/train_sam_real.py:229-234).
Tolerance: fp32 logits 1e-4-scale as in tests/test_gpu_model.py; ids / keys / untouched basket rows exact."""
import numpy as np
import pytest
import torch

from oracle import ptv2_ref as M

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _build(golden):
    import ao_amd.ptv2 as ptv2

    g, m = golden("segmentor_sam.npz"), golden("ptv2_s3dis.npz")
    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    seg = ptv2.DefaultSegmentorSAM_Image(backbone=dict(type="PT-v2m2", **cfg),
                                         criteria=[dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1)]).cuda()
    seg.backbone.load_state_dict(M.init_state(cfg, seed=int(g["state_seed"])), strict=True)
    assert all(k.startswith("backbone.") for k in seg.state_dict())  # checkpoint keys of the reference class
    batch = dict(coord=dev(m["coord"]), feat=dev(m["feat"]), offset=dev(m["offset"]), segment=dev(m["label"]),
                 scene_id=[str(s) for s in g["scene_id"]], instance=dev(g["instance"]))
    return ptv2, g, seg, batch


def test_sam_image_segmentor_and_async_basket_match_the_reference(golden):
    ptv2, g, seg, batch = _build(golden)
    seg.train()
    out, seg_dict = seg(batch)
    assert abs(float(out["loss"].detach()) - float(g["loss_train"])) < 2e-5
    keys = [str(k) for k in g["keys"]]
    assert list(seg_dict) == keys
    basket = ptv2.LogitBasket(dict(zip(keys, g["scene_points"].tolist())), 13, device="cuda")
    basket.put(seg_dict)
    out["loss"].backward()  # the step goes on while the copy runs
    basket.flush()
    for i, k in enumerate(keys):
        lg, ids = seg_dict[k]
        assert torch.equal(ids.cpu(), torch.from_numpy(g["seg_ids_%d" % i]))
        np.testing.assert_allclose(lg.cpu().numpy(), g["seg_logits_%d" % i], rtol=1e-3, atol=2e-4)
        ref = g["basket_%d" % i]
        assert np.array_equal(basket[k] == -100.0, ref == -100.0)  # exactly the rows the reference wrote
        np.testing.assert_allclose(basket[k], ref, rtol=1e-3, atol=2e-4)
        assert np.array_equal(basket[k][ids.cpu().numpy()], lg.cpu().numpy())  # bit-for-bit what the model produced
    basket.close()
    seg.backbone.load_state_dict(M.init_state(dict(M.S3DIS_CFG, drop_path_rate=0.0), seed=int(g["state_seed"])), strict=True)
    seg.eval()  # (running statistics reset, as the fixture's generator does before its eval pass)
    with torch.no_grad():
        ev = seg(batch)
        te = seg({k: v for k, v in batch.items() if k != "segment"})
    assert sorted(ev) == [str(k) for k in g["eval_keys"]] and sorted(te) == [str(k) for k in g["test_keys"]]
    assert abs(float(ev["loss"]) - float(g["loss_eval"])) < 2e-5


def test_basket_does_not_synchronise_the_step(golden):
    """put() must return before the GPU work it depends on has finished: enqueue a long kernel chain, put, and check
    the compute stream is still busy when put() returns."""
    ptv2, g, seg, batch = _build(golden)
    keys = [str(k) for k in g["keys"]]
    basket = ptv2.LogitBasket(dict(zip(keys, g["scene_points"].tolist())), 13, device="cuda", slots=4)
    seg.train()
    out, seg_dict = seg(batch)
    x = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    done = torch.cuda.Event()
    for _ in range(60):
        x = x @ x * 1e-3
    lg = {k: (v[0] + x[0, 0] * 0, v[1]) for k, v in seg_dict.items()}  # logits that depend on the chain
    basket.put(lg)
    done.record()
    assert not done.query(), "put() waited for the compute stream"
    basket.flush()
    assert done.query()
    for i, k in enumerate(keys):
        assert np.array_equal(basket[k] == -100.0, g["basket_%d" % i] == -100.0)


def test_loss_weight(golden):
    ptv2, g, seg, batch = _build(golden)
    half = ptv2.DefaultSegmentor(seg.backbone, criteria=[dict(type="CrossEntropyLoss", loss_weight=0.5, ignore_index=-1)]).eval()
    with torch.no_grad():
        assert abs(float(half(batch)["loss"]) - float(g["loss_eval_half_weight"])) < 2e-5
