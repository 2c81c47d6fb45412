"""CPU: the re-associated GVA of ao_amd/ptv2/gva.py (host logic + a torch statement of its device
stages) against the oracle's literal GroupedVectorAttention, forward and all gradients, train and
eval, including -1 neighbour slots and the running-statistics update."""
import numpy as np
import pytest
import torch

from oracle import pointops_ref as P
from oracle import ptv2_ref as M
from tests import synth
from tests.gva_torch_ref import TorchImpl
from tests.test_oracle_model import assert_grad_close, block_state


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("c,g,k", [(48, 6, 16), (96, 12, 8)])
def test_fused_host_logic_matches_oracle(mode, c, g, k):
    from ao_amd.ptv2 import gva
    from ao_amd.ptv2.model import GroupedVectorAttention

    torch.manual_seed(0)
    n = 700
    xyz = torch.from_numpy(synth.room_cloud(n, seed=3))
    off = torch.tensor([300, n], dtype=torch.int32)
    idx, _ = P.knn_query(k, xyz, off)
    idx = idx.clone()
    idx[3::11, k - 3:] = -1
    cfg = dict(M.S3DIS_CFG, patch_embed_depth=1, patch_embed_channels=c, patch_embed_groups=g, enc_depths=(),
               enc_channels=(), enc_groups=(), enc_neighbours=(), dec_depths=(), dec_channels=(), dec_groups=(),
               dec_neighbours=(), grid_sizes=(), num_classes=0)
    st = M.init_state(cfg, seed=5)
    pre = "patch_embed.blocks.blocks.0.attn."
    ast = {k_[len(pre):]: v for k_, v in st.items() if k_.startswith(pre)}
    feat0 = torch.randn(n, c)
    gout = torch.randn(n, c)

    # oracle (literal op sequence)
    ost = {"a." + k_: (v.clone().requires_grad_(True) if M.is_param(k_) else v.clone()) for k_, v in ast.items()}
    cx = M.Ctx(ost, mode == "train", update_stats=True)
    f1 = feat0.clone().requires_grad_(True)
    ref = M._gva(cx, "a", f1, xyz, idx, g)
    names = [k_ for k_ in ost if M.is_param(k_[2:])]
    rgrads = torch.autograd.grad(ref, [f1] + [ost[k_] for k_ in names], gout)

    # fused host logic on the torch stages
    mod = GroupedVectorAttention(c, g)
    mod.load_state_dict(ast, strict=True)
    mod.train(mode == "train")
    f2 = feat0.clone().requires_grad_(True)
    q, kk, v = mod.linear_q(f2), mod.linear_k(f2), mod.linear_v(f2)
    out = gva.grouped_vector_attention(mod, q, kk, v, xyz, idx, impl=TorchImpl)
    np.testing.assert_allclose(out.detach().numpy(), ref.detach().numpy(), rtol=1e-4, atol=2e-5)
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(out, [f2] + [params[k_[2:]] for k_ in names], gout)
    np.testing.assert_allclose(grads[0].numpy(), rgrads[0].numpy(), rtol=1e-3, atol=1e-4)
    for k_, gr, rg in zip(names, grads[1:], rgrads[1:]):
        if mode == "train" and k_.endswith(".0.bias"):
            continue  # exactly-zero true gradient
        assert_grad_close(gr.numpy(), rg.numpy(), k_)
    if mode == "train":
        sd = mod.state_dict()
        for k_ in ast:
            if k_.endswith(("running_mean", "running_var")):
                np.testing.assert_allclose(sd[k_].numpy(), ost["a." + k_].numpy(), rtol=1e-4, atol=1e-6, err_msg=k_)
            if k_.endswith("num_batches_tracked"):
                assert int(sd[k_]) == 1
