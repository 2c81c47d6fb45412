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


def test_attention_dropout_mask_statement_on_the_host():
    """gva.attn_drop_mask (the torch statement of ptv2_drop_factor, ao_amd/csrc/gva_common.h) on CPU: a function of
    (seed, element) only -- the same on every device and whatever the tensor shape it is viewed in --, values in
    {0, 1 / (1 - p)}, keep fraction 1 - p, and the integer hash itself against a scalar python restatement."""
    from ao_amd.ptv2 import gva

    n, k, g = 700, 16, 6
    for p in (0.1, 0.5):
        m = gva.attn_drop_mask(77, n, k, g, p, "cpu")
        assert m.shape == (n, k, g)
        vals = np.unique(m.numpy())
        assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / (1.0 - p)) < 1e-6
        assert abs(float((m > 0).float().mean()) - (1 - p)) < 0.01
        assert torch.equal(m.reshape(-1), gva.attn_drop_mask(77, n * k, 1, g, p, "cpu").reshape(-1))
        assert not torch.equal(m, gva.attn_drop_mask(78, n, k, g, p, "cpu"))
    assert bool((gva.attn_drop_mask(1, 5, 4, 3, 0.0, "cpu") == 1).all())

    def scalar(e, seed, p):
        M32 = 0xFFFFFFFF
        h = (e & M32) ^ (((e >> 32) * 0x27D4EB2F) & M32)
        h = (h * 0x9E3779B1 + seed) & M32
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & M32
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & M32
        h ^= h >> 16
        return 0.0 if h < int(p * 4294967296.0) else 1.0 / (1.0 - p)

    m = gva.attn_drop_mask(12345, 9, 4, 3, 0.3, "cpu").reshape(-1)
    for e in range(m.numel()):
        assert abs(float(m[e]) - scalar(e, 12345, 0.3)) < 1e-6, e
    seeds = [gva.next_drop_seed() for _ in range(50)]
    assert all(0 <= s < 2 ** 31 for s in seeds) and len(set(seeds)) > 45
    torch.manual_seed(3)
    a = [gva.next_drop_seed() for _ in range(4)]
    torch.manual_seed(3)
    assert a == [gva.next_drop_seed() for _ in range(4)]


def test_dropout_routing_of_the_attention_module():
    """Which shapes take the fused dropout (gva.dropout_supported) and that the module falls back to the literal op sequence
    -- not to a silently undropped fused call -- where they do not (AO_AMD_GVA=staged, k > 16)."""
    from ao_amd.ptv2 import gva

    for c, g in ((48, 6), (96, 12), (192, 24), (384, 48), (512, 64)):
        assert gva.dropout_supported(c, g, 16) and gva.dropout_supported(c, g, 8)
        assert not gva.dropout_supported(c, g, 32)
    assert not gva.dropout_supported(64, 8, 16)


def test_inverse_table_host_statement():
    """The CPU statement of the inverse neighbour table (what the device kernels of ao_amd/csrc/inverse.hip are compared with):
    lists in ascending slot order, placeholders first, every slot exactly once."""
    from ao_amd.ptv2.gva import inverse_table

    g = torch.Generator().manual_seed(1)
    n, k = 500, 6
    idx = torch.randint(-1, n, (n, k), generator=g, dtype=torch.int32)
    ptr, rows = inverse_table(idx)
    flat = idx.reshape(-1)
    assert ptr.shape == (n + 1,) and rows.shape == (n * k,) and int(ptr[n]) == n * k
    assert sorted(rows.tolist()) == list(range(n * k))
    assert torch.equal(rows[: int(ptr[0])].long(), torch.nonzero(flat == -1).reshape(-1))
    for j in (0, 1, 17, n - 1):
        start = int(ptr[j])
        end = int(ptr[j + 1]) if j + 1 <= n else n * k
        members = rows[start:end].long()
        assert torch.equal(members, torch.nonzero(flat == j).reshape(-1)), j


def test_knn_grid_plan():
    """pointops.query.KnnGrid.plan: build on first use, reuse for the same (points, offsets, stream) up to four queries, rebuild
    for other points, a larger workspace or a fifth query."""
    from ao_amd.pointops.query import KnnGrid

    xyz, off = torch.zeros(100, 3), torch.tensor([100], dtype=torch.int32)
    grid = KnnGrid()
    mode, slot, ws = grid.plan(xyz, off, 100, 100, 1, 1000, 0)
    assert (mode, slot) == (0, 0) and ws.numel() >= 1000
    for want in (1, 2, 3):
        mode, slot, ws2 = grid.plan(xyz, off, 50, 100, 1, 900, 0)
        assert (mode, slot) == (1, want) and ws2 is ws
    assert grid.plan(xyz, off, 50, 100, 1, 900, 0)[:2] == (0, 0)          # the fifth query
    assert grid.plan(xyz, off, 50, 100, 1, 900, 7)[:2] == (0, 0)          # another stream
    assert grid.plan(xyz, off, 50, 100, 1, 900, 7)[:2] == (1, 1)
    assert grid.plan(xyz.clone(), off, 50, 100, 1, 900, 7)[:2] == (0, 0)  # other points
    mode, slot, ws3 = grid.plan(xyz, off, 5000, 100, 1, 10 ** 6, 7)       # does not fit the workspace
    assert (mode, slot) == (0, 0) and ws3.numel() >= 10 ** 6
