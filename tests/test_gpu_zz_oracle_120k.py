"""The bench workloads themselves against the CPU oracle, once each (VERDICT r2 weak #1 / "do this" #4; r3 "do this" #5).

ONE 120 000-point S3DIS-cfg train step (drop_path 0) on the HIP path -- whole-model native runtime, fused attention:
`attention_bwd_point_kernel<6,48,1>` on its co-resident grid, the 24 000-workgroup `aggregate_tile_kernel<6>`, the
2 048-workgroup-capped row kernels, i.e. the size-dependent branches bench.py times -- and on `oracle/ptv2_ref.RefModule`
(torch-CPU restatement of point_transformer_v2m2_base.py + the C kNN, ~2 min on the GPU box's host threads): logits within
1e-4 absolute (north_star's bound for fp32 features), loss within 2e-5, EVERY parameter gradient in relative L2.

The file name sorts last on purpose: `pytest -x` reaches it after every other GPU test, and the two minutes of host time
are paid once."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ptv2_ref as M

pytestmark = pytest.mark.gpu

# biases in front of a training-mode BatchNorm / softmax: the true gradient is exactly zero, only negligibility is checked
ZERO_GRAD_BIAS = ("linear_q.0.bias", "linear_k.0.bias", "linear_v.bias", "linear_p_bias.0.bias", "linear_p_bias.3.bias",
                  "weight_encoding.0.bias", "weight_encoding.3.bias", "proj.0.bias", "proj_skip.0.bias", "seg_head.0.bias")


def _compare_grads(names, got, ref, rel_l2, weights):
    worst = ("", 0.0)
    for nm, a, b in zip(names, got, ref):
        if nm.endswith(ZERO_GRAD_BIAS):
            assert float(a.double().norm()) <= 2e-2 * float(weights[nm].double().norm()) + 1e-3, (nm, float(a.norm()))
            continue
        r = float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
        if r > worst[1]:
            worst = (nm, r)
        assert r < rel_l2 or float((a - b).abs().max()) < 1e-5, (nm, r, float((a - b).abs().max()))
    return worst


@pytest.mark.parametrize("tag", ["s3dis_1x120k", "scannet_2x100k"])
def test_bench_size_train_step_matches_the_cpu_oracle(tag):
    """s3dis_1x120k: bench.py's default workload (rank-0 scene).  scannet_2x100k: the ScanNet cfg at the size
    profiles/*_bench_scannet*.json times -- 4 stages, C up to 512 / G = 64, K = 8 patch embedding, "map" unpooling, two
    clouds per batch: the instantiations no S3DIS run reaches (VERDICT r3 "do this" #5)."""
    import ao_amd.ptv2 as ptv2
    from ao_amd import synth

    if tag == "s3dis_1x120k":
        cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
        b = synth.scene_batch([0], point_max=120000, room=1)  # bench.py's rank-0 scene
        want_points = 120000
    else:
        cfg = dict(M.SCANNET_CFG, drop_path_rate=0.0)
        b = synth.scene_batch([0, 1], point_max=100000, in_channels=cfg["in_channels"], num_classes=cfg["num_classes"], room=1)
        want_points = 200000
    cpu = {k: torch.from_numpy(v) for k, v in b.items()}
    gpu = {k: v.cuda() for k, v in cpu.items()}
    assert cpu["coord"].shape[0] == want_points
    state = M.init_state(cfg, seed=31)
    model = ptv2.PointTransformerV2(**cfg).cuda().train()
    model.load_state_dict(state, strict=True)
    logits = model(gpu)
    loss = F.cross_entropy(logits, gpu["segment"], ignore_index=-1)
    names = [nm for nm, _ in model.named_parameters()]
    grads = [g.cpu() for g in torch.autograd.grad(loss, list(model.parameters()))]
    logits, loss = logits.detach().cpu(), float(loss.detach())
    del model
    torch.cuda.empty_cache()

    ref = M.RefModule(cfg, seed=31, randomize_bn=True).train()
    ref_logits = ref(cpu)
    ref_loss = F.cross_entropy(ref_logits, cpu["segment"], ignore_index=-1)
    ref_params = dict(ref.named_parameters())
    ref_grads = torch.autograd.grad(ref_loss, [ref_params[nm.replace(".", "/")] for nm in names])
    np.testing.assert_allclose(logits.numpy(), ref_logits.detach().numpy(), rtol=0, atol=1e-4)
    assert abs(loss - float(ref_loss.detach())) < 2e-5
    weights = {nm: g for nm, g in zip(names, ref_grads)}
    weights = {nm: weights.get(nm[:-4] + "weight", weights[nm]) for nm in names}
    worst = _compare_grads(names, grads, list(ref_grads), 2e-2, weights)
    print("oracle %s: max |dlogit| %.2e, loss %.6f / %.6f, worst gradient %s rel L2 %.2e"
          % (tag, float((logits - ref_logits.detach()).abs().max()), loss, float(ref_loss.detach()), *worst))
