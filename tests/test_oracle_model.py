"""CPU: the functional PT-v2m2 oracle (oracle/ptv2_ref.py) against fixtures produced by the
reference nn.Module itself (tests/golden/make_golden.py) -- this is what pins rows a9-a14."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ptv2_ref as M
from tests.conftest import GOLDEN


def t(a):
    return torch.from_numpy(np.asarray(a))


def assert_grad_close(a, b, name="", rel_l2=5e-3, frac_max=2e-2):
    """Parameter gradients sum over ~1e4-1e6 rows through ReLU/BN: a 1e-6 input difference can flip a
    ReLU and move one row of a gradient by a whole term, so compare in relative L2 plus a loose
    per-element bound (fraction of the tensor's max)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-12)
    assert np.linalg.norm(a - b) <= rel_l2 * max(np.linalg.norm(b), 1e-12) + 1e-4, (name, np.linalg.norm(a - b), np.linalg.norm(b))
    assert np.abs(a - b).max() <= frac_max * scale + 1e-4, (name, np.abs(a - b).max(), scale)


def digest(state):
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(state[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def block_state(seed):
    cfg = dict(M.S3DIS_CFG, patch_embed_depth=1, enc_depths=(), enc_channels=(), enc_groups=(), enc_neighbours=(),
               dec_depths=(), dec_channels=(), dec_groups=(), dec_neighbours=(), grid_sizes=(), num_classes=0)
    st = M.init_state(cfg, seed=seed)
    pre = "patch_embed.blocks.blocks.0."
    return {k[len(pre):]: v for k, v in st.items() if k.startswith(pre)}


def test_state_manifest_matches_reference_module():
    man = json.load(open(os.path.join(GOLDEN, "state_manifest.json")))
    for tag, cfg in (("s3dis", M.S3DIS_CFG), ("scannet", M.SCANNET_CFG)):
        st = M.init_state(cfg, seed=0)
        assert {k: list(v.shape) for k, v in st.items()} == man[tag]
        assert sum(v.numel() for k, v in st.items() if M.is_param(k)) == man[tag + "_num_params"]
    assert len(man["s3dis"]) == 840 and man["s3dis_num_params"] == 3908641


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_gva_and_block_match_reference_module(golden, mode):
    g = golden("gva_block.npz")
    bst = block_state(int(g["state_seed"]))
    assert digest(bst) == str(g["digest"]), "torch CPU RNG stream changed: regenerate goldens"
    st = {"b." + k: (v.clone().requires_grad_(True) if M.is_param(k) else v.clone()) for k, v in bst.items()}
    cx = M.Ctx(st, mode == "train", update_stats=True)
    feat = t(g["feat"]).requires_grad_(True)
    xyz, idx = t(g["xyz"]), t(g["idx"])
    out = M._gva(cx, "b.attn", feat, xyz, idx, 6)
    np.testing.assert_allclose(out.detach().numpy(), g["attn_out_" + mode], rtol=1e-4, atol=1e-4)
    names = [k for k in st if k.startswith("b.attn.") and M.is_param(k[2:])]
    grads = torch.autograd.grad(out, [feat] + [st[k] for k in names], t(g["attn_gout"]))
    np.testing.assert_allclose(grads[0].numpy(), g["attn_gfeat_" + mode], rtol=1e-3, atol=1e-4)
    for k, gr in zip(names, grads[1:]):
        ref = g["attn_g_%s_%s" % (mode, k[len("b.attn."):])]
        if mode == "train" and k.endswith(".0.bias"):
            # a bias feeding a training-mode BatchNorm has an exactly-zero true gradient; both sides hold
            # only rounding noise (amplified by rstd), so check it is negligible next to the weight's grad
            wref = g["attn_g_%s_%s" % (mode, k[len("b.attn."):-4] + "weight")]
            assert np.linalg.norm(gr.numpy()) <= 2e-3 * np.linalg.norm(wref) + 1e-4, k
            continue
        assert_grad_close(gr.numpy(), ref, k)

    st = {"b." + k: (v.clone().requires_grad_(True) if M.is_param(k) else v.clone()) for k, v in bst.items()}
    cx = M.Ctx(st, mode == "train", update_stats=True)
    y = M._block(cx, "b", feat, xyz, idx, 6)
    np.testing.assert_allclose(y.detach().numpy(), g["block_out_" + mode], rtol=1e-4, atol=1e-4)
    (gf,) = torch.autograd.grad(y, feat, t(g["attn_gout"]))
    np.testing.assert_allclose(gf.numpy(), g["block_gfeat_" + mode], rtol=1e-3, atol=1e-4)
    if mode == "train":
        for k in g.files:
            if k.startswith("block_buf_"):
                np.testing.assert_allclose(st["b." + k[len("block_buf_"):]].numpy(), g[k], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag,cfg", [("s3dis", M.S3DIS_CFG), ("scannet", M.SCANNET_CFG)])
def test_full_model_matches_reference_module(golden, tag, cfg):
    g = golden("ptv2_%s.npz" % tag)
    cfg = dict(cfg, drop_path_rate=0.0)
    st0 = M.init_state(cfg, seed=int(g["state_seed"]))
    assert digest(st0) == str(g["digest"])
    coord, feat, offset, label = t(g["coord"]), t(g["feat"]), t(g["offset"]), t(g["label"])
    for mode in ("train", "eval"):
        st = {k: (v.clone().requires_grad_(True) if M.is_param(k) else v.clone()) for k, v in st0.items()}
        logits = M.forward(st, cfg, coord, feat, offset, training=(mode == "train"))
        np.testing.assert_allclose(logits.detach().numpy(), g["logits_" + mode], rtol=1e-3, atol=1e-4)
        loss = F.cross_entropy(logits, label, ignore_index=-1)
        assert abs(float(loss) - float(g["loss_" + mode])) < 1e-5
        if mode == "train":
            watch = [k[len("grad_"):] for k in g.files if k.startswith("grad_")]
            grads = torch.autograd.grad(loss, [st[w] for w in watch])
            for w, gr in zip(watch, grads):
                ref = g["grad_" + w]
                assert_grad_close(gr.numpy(), ref, w)
