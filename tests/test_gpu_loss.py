"""GPU: the fused softmax cross-entropy (ao_amd/csrc/loss.hip) against torch.nn.functional.cross_entropy."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c", [(120000, 13), (4001, 20), (257, 13), (1, 5)])
@pytest.mark.parametrize("ignore_frac", [0.0, 0.1, 0.9])
def test_cross_entropy_matches_torch(n, c, ignore_frac):
    from ao_amd.ptv2.segmentor import cross_entropy

    torch.manual_seed(n + c)
    logits = (torch.randn(n, c, device="cuda") * 3).requires_grad_(True)
    label = torch.randint(0, c, (n,), device="cuda")
    label[torch.rand(n, device="cuda") < ignore_frac] = -1
    if n > 1:
        label[0] = 2  # at least one labelled point
    ref = F.cross_entropy(logits, label, ignore_index=-1)
    (g_ref,) = torch.autograd.grad(ref * 1.7, [logits])
    out = cross_entropy(logits, label, -1)
    (g,) = torch.autograd.grad(out * 1.7, [logits])
    if torch.isnan(ref):
        assert torch.isnan(out)
        return
    np.testing.assert_allclose(float(out.detach()), float(ref.detach()), rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(g.cpu().numpy(), g_ref.cpu().numpy(), rtol=1e-4, atol=1e-8)


def test_segmentor_loss_uses_the_kernel_and_matches():
    import ao_amd.ptv2 as ptv2
    from ao_amd import synth

    torch.manual_seed(0)
    seg = ptv2.DefaultSegmentor(dict(ptv2.S3DIS_BACKBONE, drop_path_rate=0.0)).cuda().train()
    b = synth.scene_batch([0], point_max=3000, room=1)
    data = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    logits = seg.backbone(data)
    np.testing.assert_allclose(float(seg.loss(logits, data["segment"]).detach()),
                               float(F.cross_entropy(logits, data["segment"], ignore_index=-1).detach()), rtol=2e-6)


def test_out_of_range_label_poisons_the_loss(monkeypatch):
    """torch device-asserts on a label outside [0, C) that is not ignore_index; the HIP kernel must not
    silently treat it as ignored: the loss becomes NaN (no synchronisation), AO_AMD_CHECK_LABELS=1 raises."""
    from ao_amd.ptv2.segmentor import cross_entropy

    torch.manual_seed(1)
    logits = torch.randn(5000, 13, device="cuda", requires_grad=True)
    label = torch.randint(0, 13, (5000,), device="cuda")
    label[::7] = -1
    assert torch.isfinite(cross_entropy(logits, label, -1))
    label[1234] = 255  # e.g. a dataset whose "unlabelled" value is not the configured ignore_index
    assert torch.isnan(cross_entropy(logits, label, -1))
    clean = torch.where(label == -1, torch.zeros_like(label), label)  # only 255 left as a non-class value
    assert torch.isfinite(cross_entropy(logits, clean, 255))  # fine when that IS the ignore_index
    monkeypatch.setenv("AO_AMD_CHECK_LABELS", "1")
    with pytest.raises(ValueError, match="1 labels are neither"):
        cross_entropy(logits, clean, -1)


def test_no_labelled_point_gives_nan_loss_and_zero_gradient():
    from ao_amd.ptv2.segmentor import cross_entropy

    logits = torch.randn(300, 13, device="cuda", requires_grad=True)
    label = torch.full((300,), -1, device="cuda")
    out = cross_entropy(logits, label, -1)
    assert torch.isnan(out)  # 0 / 0, as torch
    (g,) = torch.autograd.grad(out, [logits])
    assert torch.equal(g, torch.zeros_like(g))  # torch's CPU kernel: zeros
