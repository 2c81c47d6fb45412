"""The stage modules driven through the reference's OWN forward signatures
(point_transformer_v2m2_base.py:244 GridPool(points, start), :305 UnpoolWithSkip(points, skip_points, cluster),
:356 Encoder(points), :400 Decoder(points, skip_points, cluster), :441 GVAPatchEmbed(points)), wired exactly as the
reference's PointTransformerV2.forward wires them (:556-576), against the fixtures captured from the reference
nn.Module (tests/golden/ptv2_*.npz).  Tolerance: fp32 features 1e-4 (north_star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ptv2_ref as M
from tests.test_oracle_model import assert_grad_close, digest

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def reference_wiring(model, data_dict):
    """Literal restatement of the reference's top-level forward (:556-576) over the stage modules' call API."""
    coord, feat, offset = data_dict["coord"], data_dict["feat"], data_dict["offset"].int()
    points = [coord, feat, offset]
    points = model.patch_embed(points)
    skips = [[points]]
    for i in range(model.num_stages):
        points, cluster = model.enc_stages[i](points)
        skips[-1].append(cluster)
        skips.append([points])
    points = skips.pop(-1)[0]
    for i in reversed(range(model.num_stages)):
        skip_points, cluster = skips.pop(-1)
        points = model.dec_stages[i](points, skip_points, cluster)
    coord, feat, offset = points
    return model.seg_head(feat)


@pytest.mark.parametrize("tag", ["s3dis", "scannet"])
def test_stage_call_api_reproduces_the_reference_module(golden, tag):
    import ao_amd.ptv2 as ptv2

    g = golden("ptv2_%s.npz" % tag)
    cfg = dict(M.S3DIS_CFG if tag == "s3dis" else M.SCANNET_CFG, drop_path_rate=0.0)
    st0 = M.init_state(cfg, seed=int(g["state_seed"]))
    assert digest(st0) == str(g["digest"])
    model = ptv2.PointTransformerV2(**cfg).cuda()
    data = dict(coord=dev(g["coord"]), feat=dev(g["feat"]), offset=dev(g["offset"]))
    label = dev(g["label"])
    for mode in ("train", "eval"):
        model.load_state_dict(st0, strict=True)
        model.train(mode == "train")
        logits = reference_wiring(model, data)
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits_" + mode], rtol=1e-3, atol=2e-4)
        loss = F.cross_entropy(logits, label, ignore_index=-1)
        assert abs(float(loss) - float(g["loss_" + mode])) < 2e-5
        if mode == "train":
            watch = [k[len("grad_"):] for k in g.files if k.startswith("grad_")]
            params = dict(model.named_parameters())
            grads = torch.autograd.grad(loss, [params[w] for w in watch])
            for w, gr in zip(watch, grads):
                assert_grad_close(gr.cpu().numpy(), g["grad_" + w], w, rel_l2=3e-2, frac_max=5e-2)
        # and the geometry-plan forward of the same module is the same computation
        model.load_state_dict(st0, strict=True)
        with torch.no_grad():
            planned = model(data)
        # (the same kernels in both paths but for the BatchNorm statistics of the Linear + BatchNorm layers between the Blocks --
        # the planned forward merges the GEMM's 64-row records, the stage modules run bn_stats -- and the grouping of the unpool's
        # add and the fold launches: a few ulp at |logit| ~ 2 -- 7.1e-6 observed on 8 of 52 000 elements of the S3DIS
        # configuration, 2.4e-5 on 7 of 60 000 of the five-stage ScanNet one; the bound against the golden logits above is 2e-4)
        np.testing.assert_allclose(planned.cpu().numpy(), logits.detach().cpu().numpy(), rtol=0, atol=5e-5)


def test_stage_return_contracts():
    """Shapes / types each stage hands to the next, as the reference documents them."""
    import ao_amd.ptv2 as ptv2
    from ao_amd import synth

    b = synth.scene_batch([1, 2], point_max=4000)
    coord, feat, offset = (torch.from_numpy(b[k]).cuda() for k in ("coord", "feat", "offset"))
    torch.manual_seed(0)
    pe = ptv2.model.GVAPatchEmbed(1, 6, 48, 6, 8).cuda()
    enc = ptv2.model.Encoder(1, 48, 96, 12, grid_size=0.1).cuda()
    dec_map = ptv2.model.Decoder(96, 48, 48, 6, 1, unpool_backend="map").cuda()
    dec_int = ptv2.model.Decoder(96, 48, 48, 6, 1, unpool_backend="interp").cuda()
    p0 = pe([coord, feat, offset.int()])
    assert p0[0] is coord and p0[1].shape == (coord.shape[0], 48)
    p1, cluster = enc(p0)
    assert cluster.dtype == torch.int64 and cluster.shape == (coord.shape[0],)
    n1 = p1[0].shape[0]
    assert int(cluster.max()) == n1 - 1 and p1[1].shape == (n1, 96) and int(p1[2][-1]) == n1
    # explicit `start` = the per-cloud minimum reproduces the default clustering
    from ao_amd.ptv2.geometry import segment_minmax

    (q_coord, q_feat, q_off), q_cluster = enc.down(p0, start=segment_minmax(coord, offset.int())[0])
    assert torch.equal(q_cluster, cluster) and torch.equal(q_off, p1[2])
    np.testing.assert_allclose(q_coord.cpu().numpy(), p1[0].cpu().numpy(), rtol=0, atol=1e-6)
    for dec in (dec_map, dec_int):
        out = dec(p1, p0, cluster)
        assert out[0] is coord and out[1].shape == (coord.shape[0], 48) and out[2] is p0[2]
        out[1].sum().backward(retain_graph=True)  # both unpool backwards run (ordered per-cluster sum / inverse-table gather)
    assert all(p.grad is not None for p in dec_int.parameters())
