"""tests/golden/make_golden.py -- regenerates the golden fixtures in this directory.

Runs ONLY in the build container (it imports the reference from /root/reference,
which does not exist on the GPU box).  Nothing here is read at test time except
the .npz / .json files it writes.

What is executed from the reference, unmodified, on CPU torch:
  * libs/pointops/functions/*.py  (the `pointops` Python API: autograd wrappers,
    pure-torch `grouping` / `interpolation`, query_and_group, ...)
  * pointcept/models/point_transformer_v2/point_transformer_v2m2_base.py
  * pointcept/models/utils.py
What is stubbed (the reference has no CPU implementation of these):
  * pointops._C          -> oracle/pointops_oracle.c through ctypes (so fixtures that
                            involve a CUDA kernel are *oracle* outputs, "parity unpinned")
  * torch.cuda.{Float,Int}Tensor -> CPU constructors (the wrappers hard-code them)
  * torch_geometric.nn.pool.voxel_grid, torch_scatter.segment_csr -> oracle/ptv2_ref.py
  * timm.models.layers.DropPath -> identity (fixtures use drop_path_rate=0)
  * pointcept.models.builder.MODELS -> no-op registry

usage:  python tests/golden/make_golden.py
"""
import ctypes
import hashlib
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import pointops_ref as OP  # noqa: E402
from oracle import ptv2_ref as OM  # noqa: E402
from tests import synth  # noqa: E402


# ------------------------------------------------------------------ stubs --
def _fp(t):
    assert t.dtype == torch.float32 and t.is_contiguous()
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))


def _ip(t):
    assert t.dtype == torch.int32 and t.is_contiguous()
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_int))


def install_stubs():
    L = OP.lib()
    C = types.ModuleType("pointops._C")
    C.knn_query_cuda = lambda m, ns, xyz, nxyz, off, noff, idx, d2: L.oracle_knn_query(
        m, ns, _fp(xyz), _fp(nxyz), _ip(off), _ip(noff), _ip(idx), _fp(d2), 0)
    C.farthest_point_sampling_cuda = lambda b, nmax, xyz, off, noff, tmp, idx: L.oracle_farthest_point_sampling(
        b, int(nmax), _fp(xyz), _ip(off), _ip(noff), _fp(tmp), _ip(idx))
    C.grouping_forward_cuda = lambda m, ns, c, inp, idx, out: L.oracle_grouping_forward(
        m, ns, c, _fp(inp), _ip(idx), _fp(out))
    C.grouping_backward_cuda = lambda m, ns, c, go, idx, gi: L.oracle_grouping_backward(
        m, ns, c, _fp(go.contiguous()), _ip(idx), _fp(gi))
    C.interpolation_forward_cuda = lambda n, c, k, inp, idx, w, out: L.oracle_interpolation_forward(
        n, c, k, _fp(inp), _ip(idx), _fp(w), _fp(out))
    C.interpolation_backward_cuda = lambda n, c, k, go, idx, w, gi: L.oracle_interpolation_backward(
        n, c, k, _fp(go.contiguous()), _ip(idx), _fp(w), _fp(gi))
    C.subtraction_forward_cuda = lambda n, ns, c, a, b, idx, out: L.oracle_subtraction_forward(
        n, ns, c, _fp(a), _fp(b), _ip(idx), _fp(out))
    C.subtraction_backward_cuda = lambda n, ns, c, idx, go, g1, g2: L.oracle_subtraction_backward(
        n, ns, c, _ip(idx), _fp(go.contiguous()), _fp(g1), _fp(g2))
    C.aggregation_forward_cuda = lambda n, ns, c, wc, inp, pos, w, idx, out: L.oracle_aggregation_forward(
        n, ns, c, wc, _fp(inp), _fp(pos), _fp(w), _ip(idx), _fp(out))
    C.aggregation_backward_cuda = lambda n, ns, c, wc, inp, pos, w, idx, go, gi, gp, gw: L.oracle_aggregation_backward(
        n, ns, c, wc, _fp(inp), _fp(pos), _fp(w), _ip(idx), _fp(go.contiguous()), _fp(gi), _fp(gp), _fp(gw))
    C.attention_relation_step_forward_cuda = lambda m, g, c, q, k, w, t, r, out: L.oracle_attention_relation_step_forward(
        m, g, c, _fp(q), _fp(k), _fp(w), _ip(t), _ip(r), _fp(out))
    C.attention_relation_step_backward_cuda = lambda m, g, c, q, gq, k, gk, w, gw, t, r, go: L.oracle_attention_relation_step_backward(
        m, g, c, _fp(q), _fp(gq), _fp(k), _fp(gk), _fp(w), _fp(gw), _ip(t), _ip(r), _fp(go.contiguous()))
    C.attention_fusion_step_forward_cuda = lambda m, g, c, w, v, t, r, out: L.oracle_attention_fusion_step_forward(
        m, g, c, _fp(w), _fp(v), _ip(t), _ip(r), _fp(out))
    C.attention_fusion_step_backward_cuda = lambda m, g, c, w, gw, v, gv, t, r, go: L.oracle_attention_fusion_step_backward(
        m, g, c, _fp(w), _fp(gw), _fp(v), _fp(gv), _ip(t), _ip(r), _fp(go.contiguous()))
    C.ball_query_cuda = C.random_ball_query_cuda = None
    sys.modules["pointops._C"] = C

    torch.cuda.FloatTensor = lambda *s: torch.empty(*s, dtype=torch.float32) if s else torch.empty(0)
    torch.cuda.IntTensor = lambda *s: torch.empty(*s, dtype=torch.int32)

    # `import pointops` -> the reference's python package.  libs/pointops/setup.py:20-28 installs
    # the directory functions/ AS the package `pointops` (package_dir={"pointops": "functions"}).
    fdir = os.path.join(REF, "libs/pointops/functions")
    spec = importlib.util.spec_from_file_location(
        "pointops", os.path.join(fdir, "__init__.py"), submodule_search_locations=[fdir])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["pointops"] = pkg
    spec.loader.exec_module(pkg)

    timm_layers = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            assert p == 0.0, "fixtures are generated with drop_path_rate=0"

        def forward(self, x):
            return x

    timm_layers.DropPath = DropPath
    for name in ("timm", "timm.models"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["timm.models.layers"] = timm_layers

    tg_pool = types.ModuleType("torch_geometric.nn.pool")
    tg_pool.voxel_grid = lambda pos, size, batch, start=0: OM.voxel_grid(pos, size, batch, start)
    for name in ("torch_geometric", "torch_geometric.nn"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torch_geometric.nn.pool"] = tg_pool

    ts = types.ModuleType("torch_scatter")
    ts.segment_csr = lambda src, indptr, reduce="sum": OM.segment_csr(src, indptr, reduce)
    sys.modules["torch_scatter"] = ts

    builder = types.ModuleType("pointcept.models.builder")

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls

    builder.MODELS = _Reg()
    for name in ("pointcept", "pointcept.models"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pointcept.models.builder"] = builder
    spec = importlib.util.spec_from_file_location(
        "pointcept.models.utils", os.path.join(REF, "pointcept/models/utils.py"))
    utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(utils)
    sys.modules["pointcept.models.utils"] = utils


def load_reference_model_module():
    spec = importlib.util.spec_from_file_location(
        "ref_ptv2m2", os.path.join(REF, "pointcept/models/point_transformer_v2/point_transformer_v2m2_base.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def state_digest(state):
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(state[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def save(name, **arrays):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items()})


# --------------------------------------------------------------- fixtures --
def gen_ops(pointops):
    torch.manual_seed(0)
    # -- knn: random 2-cloud, lattice (ties), short segment (< k) ------------
    xyz = torch.from_numpy(synth.random_cloud(2048, seed=1))
    offset = torch.tensor([900, 2048], dtype=torch.int32)
    out = dict(xyz=xyz, offset=offset)
    for k in (1, 3, 8, 16):
        idx, dist = pointops.knn_query(k, xyz, offset)
        out["idx_k%d" % k], out["dist_k%d" % k] = idx, dist
    new_xyz = torch.from_numpy(synth.random_cloud(700, seed=2))
    new_offset = torch.tensor([300, 700], dtype=torch.int32)
    idx, dist = pointops.knn_query(3, xyz, offset, new_xyz, new_offset)
    out.update(new_xyz=new_xyz, new_offset=new_offset, cross_idx_k3=idx, cross_dist_k3=dist)
    save("knn_random.npz", **out)

    lat = torch.from_numpy(synth.lattice_cloud(8, 8, 6, 0.25))  # 384 pts, massive ties
    loff = torch.tensor([200, 384], dtype=torch.int32)
    out = dict(xyz=lat, offset=loff)
    for k in (4, 16):
        idx, dist = pointops.knn_query(k, lat, loff)
        out["idx_k%d" % k], out["dist_k%d" % k] = idx, dist
    save("knn_lattice.npz", **out)

    sxyz = torch.from_numpy(synth.random_cloud(40, seed=3))
    soff = torch.tensor([5, 12, 40], dtype=torch.int32)  # segments of 5, 7, 28 points; k=16
    idx, dist = pointops.knn_query(16, sxyz, soff)
    save("knn_short.npz", xyz=sxyz, offset=soff, idx_k16=idx, dist_k16=dist)

    # -- FPS -------------------------------------------------------------------
    fx = torch.from_numpy(synth.random_cloud(4096, seed=4))
    foff = torch.tensor([1500, 4096], dtype=torch.int32)
    fnoff = torch.tensor([375, 1024], dtype=torch.int32)
    fidx = pointops.farthest_point_sampling(fx, foff, fnoff)
    dup = torch.from_numpy(synth.lattice_cloud(6, 6, 4, 0.5)).repeat(2, 1).contiguous()  # duplicates + ties
    doff = torch.tensor([dup.shape[0]], dtype=torch.int32)
    dnoff = torch.tensor([100], dtype=torch.int32)
    didx = pointops.farthest_point_sampling(dup, doff, dnoff)
    save("fps.npz", xyz=fx, offset=foff, new_offset=fnoff, idx=fidx,
         dup_xyz=dup, dup_offset=doff, dup_new_offset=dnoff, dup_idx=didx)

    # -- gather-style ops, fwd + bwd (n=160, k=16, c=48) -----------------------
    n, k, c, g = 160, 16, 48, 6
    pts = torch.from_numpy(synth.random_cloud(n, seed=5))
    off = torch.tensor([60, n], dtype=torch.int32)
    idx, _ = pointops.knn_query(k, pts, off)
    idx_m1 = idx.clone()
    idx_m1[::7, 11:] = -1  # placeholder slots as produced for short segments
    feat = torch.randn(n, c, requires_grad=True)

    o = pointops.grouping(idx_m1, feat, pts, with_xyz=True)  # reference pure-torch path
    go = torch.randn_like(o)
    (gfeat,) = torch.autograd.grad(o, feat, go)
    o2 = pointops.grouping2(feat, idx)
    go2 = torch.randn_like(o2)
    (gfeat2,) = torch.autograd.grad(o2, feat, go2)
    save("grouping.npz", xyz=pts, offset=off, idx=idx, idx_m1=idx_m1, feat=feat, out_xyz=o, grad_out=go,
         grad_feat=gfeat, out2=o2, grad_out2=go2, grad_feat2=gfeat2)

    coarse = torch.from_numpy(synth.random_cloud(50, seed=6))
    coff = torch.tensor([20, 50], dtype=torch.int32)
    cfeat = torch.randn(50, c, requires_grad=True)
    it = pointops.interpolation(coarse, pts, cfeat, coff, off)  # reference pure-torch path
    git = torch.randn_like(it)
    (gcf,) = torch.autograd.grad(it, cfeat, git)
    it2 = pointops.interpolation2(coarse, pts, cfeat, coff, off)
    (gcf2,) = torch.autograd.grad(it2, cfeat, git)
    save("interpolation.npz", xyz=coarse, offset=coff, new_xyz=pts, new_offset=off, feat=cfeat, out=it,
         grad_out=git, grad_feat=gcf, out2=it2, grad_feat2=gcf2)

    a = torch.randn(n, c, requires_grad=True)
    b = torch.randn(n, c, requires_grad=True)
    s = pointops.subtraction(a, b, idx)
    gs = torch.randn_like(s)
    ga, gb = torch.autograd.grad(s, (a, b), gs)
    save("subtraction.npz", idx=idx, in1=a, in2=b, out=s, grad_out=gs, grad_in1=ga, grad_in2=gb)

    wc = 8
    inp = torch.randn(n, c, requires_grad=True)
    pos = torch.randn(n, k, c, requires_grad=True)
    w = torch.randn(n, k, wc, requires_grad=True)
    ag = pointops.aggregation(inp, pos, w, idx)
    gag = torch.randn_like(ag)
    gi, gp, gw = torch.autograd.grad(ag, (inp, pos, w), gag)
    save("aggregation.npz", idx=idx, input=inp, position=pos, weight=w, out=ag, grad_out=gag,
         grad_input=gi, grad_position=gp, grad_weight=gw)

    ci = c // g
    q = torch.randn(n, g, ci, requires_grad=True)
    kk = torch.randn(n, g, ci, requires_grad=True)
    v = torch.randn(n, g, ci, requires_grad=True)
    wvec = torch.rand(ci)
    tgt = torch.arange(n, dtype=torch.int32).repeat_interleave(k)
    ref = idx.reshape(-1).contiguous()
    rel = pointops.attention_relation_step(q, kk, wvec, tgt, ref)
    grel = torch.randn_like(rel)
    gq, gk = torch.autograd.grad(rel, (q, kk), grel)
    aw = torch.softmax(rel.view(n, k, g), 1).reshape(n * k, g).detach().requires_grad_(True)
    fu = pointops.attention_fusion_step(aw, v, tgt, ref)
    gfu = torch.randn_like(fu)
    gaw, gv = torch.autograd.grad(fu, (aw, v), gfu)
    save("attention.npz", query=q, key=kk, value=v, weight=wvec, index_target=tgt, index_refer=ref,
         relation=rel, grad_relation=grel, grad_query=gq, grad_key=gk, attn=aw, fused=fu, grad_fused=gfu,
         grad_attn=gaw, grad_value=gv)


def gen_gva(ref):
    """One GroupedVectorAttention and one Block of the reference, train & eval."""
    n, c, g, k = 1024, 48, 6, 16
    pts = torch.from_numpy(synth.room_cloud(n, seed=7))
    off = torch.tensor([400, n], dtype=torch.int32)
    torch.manual_seed(11)
    blk = ref.Block(embed_channels=c, groups=g)
    st = OM.init_state(dict(OM.S3DIS_CFG, patch_embed_depth=1, enc_depths=(), enc_channels=(), enc_groups=(),
                            enc_neighbours=(), dec_depths=(), dec_channels=(), dec_groups=(), dec_neighbours=(),
                            grid_sizes=(), num_classes=0), seed=21)
    pre = "patch_embed.blocks.blocks.0."
    bst = {k_[len(pre):]: v for k_, v in st.items() if k_.startswith(pre)}
    blk.load_state_dict(bst, strict=True)
    import pointops
    idx, _ = pointops.knn_query(k, pts, off)
    idx_m1 = idx.clone()
    idx_m1[5::9, 13:] = -1
    feat0 = torch.randn(n, c)
    out = dict(xyz=pts, offset=off, idx=idx_m1, feat=feat0, state_seed=21, digest=state_digest(bst))
    for mode in ("train", "eval"):
        blk.train(mode == "train")
        blk.load_state_dict(bst, strict=True)  # reset running stats
        feat = feat0.clone().requires_grad_(True)
        a = blk.attn(feat, pts, idx_m1)
        ga = torch.randn(n, c, generator=torch.Generator().manual_seed(5))
        grads = torch.autograd.grad(a, [feat] + [p for p in blk.attn.parameters()], ga)
        out["attn_out_" + mode] = a
        out["attn_gout"] = ga
        out["attn_gfeat_" + mode] = grads[0]
        for (nm, _), gr in zip(blk.attn.named_parameters(), grads[1:]):
            out["attn_g_%s_%s" % (mode, nm)] = gr
        blk.load_state_dict(bst, strict=True)
        feat = feat0.clone().requires_grad_(True)
        _, y, _ = blk([pts, feat, off], idx_m1)
        grads = torch.autograd.grad(y, [feat] + [p for p in blk.parameters()], ga)
        out["block_out_" + mode] = y
        out["block_gfeat_" + mode] = grads[0]
        for (nm, _), gr in zip(blk.named_parameters(), grads[1:]):
            if nm in ("fc1.weight", "fc3.weight", "norm2.norm.weight", "attn.linear_v.weight",
                      "attn.linear_p_bias.0.weight", "attn.weight_encoding.3.bias"):
                out["block_g_%s_%s" % (mode, nm)] = gr
        if mode == "train":  # running stats after one training forward
            for nm, b_ in blk.named_buffers():
                if nm.endswith("running_mean") or nm.endswith("running_var"):
                    out["block_buf_" + nm] = b_.clone()
    save("gva_block.npz", **out)


def gen_gva_pe(ref):
    """GroupedVectorAttention with the two positional-encoding switches away from the configs' values
    (point_transformer_v2m2_base.py:80-86,113-115): (pe_multiplier, pe_bias) = (True, True) and (False, False), train & eval,
    the module's own initialisation (the state travels in the fixture: the extra Sequential has no slot in init_state)."""
    n, c, g, k = 512, 48, 6, 16
    pts = torch.from_numpy(synth.room_cloud(n, seed=9))
    off = torch.tensor([200, n], dtype=torch.int32)
    import pointops
    idx, _ = pointops.knn_query(k, pts, off)
    idx[3::11, 14:] = -1
    feat0 = torch.randn(n, c, generator=torch.Generator().manual_seed(2))
    ga = torch.randn(n, c, generator=torch.Generator().manual_seed(6))
    out = dict(xyz=pts, offset=off, idx=idx, feat=feat0, gout=ga)
    for tag, (mult, bias) in (("mult_bias", (True, True)), ("plain", (False, False))):
        torch.manual_seed(31)
        attn = ref.GroupedVectorAttention(embed_channels=c, groups=g, pe_multiplier=mult, pe_bias=bias)
        with torch.no_grad():  # BatchNorm affine / running statistics away from their defaults
            gen = torch.Generator().manual_seed(32)
            for nm, p_ in attn.named_parameters():
                if "norm" in nm or nm.endswith(".1.norm.weight") or nm.endswith(".1.norm.bias"):
                    p_.add_(0.2 * torch.randn(p_.shape, generator=gen))
            for nm, b_ in attn.named_buffers():
                if nm.endswith("running_mean"):
                    b_.copy_(0.3 * torch.randn(b_.shape, generator=gen))
                if nm.endswith("running_var"):
                    b_.copy_(0.5 + torch.rand(b_.shape, generator=gen))
        st = {k_: v.clone() for k_, v in attn.state_dict().items()}
        for k_, v in st.items():
            out["%s_state_%s" % (tag, k_)] = v
        for mode in ("train", "eval"):
            attn.load_state_dict(st, strict=True)
            attn.train(mode == "train")
            feat = feat0.clone().requires_grad_(True)
            a = attn(feat, pts, idx)
            grads = torch.autograd.grad(a, [feat] + list(attn.parameters()), ga)
            out["%s_out_%s" % (tag, mode)] = a
            out["%s_gfeat_%s" % (tag, mode)] = grads[0]
            for (nm, _), gr in zip(attn.named_parameters(), grads[1:]):
                out["%s_g_%s_%s" % (tag, mode, nm)] = gr
            if mode == "train":
                for nm, b_ in attn.named_buffers():
                    if nm.endswith("running_mean") or nm.endswith("running_var"):
                        out["%s_buf_%s" % (tag, nm)] = b_.clone()
    save("gva_pe.npz", **out)


def gen_model(ref):
    """Full PT-v2m2, S3DIS config (drop_path 0) on 2 small room clouds; plus a 4-stage / 'map' variant."""
    manifest = {}
    for tag, base, n_pts, seed in (("s3dis", OM.S3DIS_CFG, 4000, 31), ("scannet", OM.SCANNET_CFG, 3000, 32)):
        cfg = dict(base, drop_path_rate=0.0)
        model = ref.PointTransformerV2(**cfg)
        manifest[tag] = {k: list(v.shape) for k, v in model.state_dict().items()}
        manifest[tag + "_num_params"] = sum(p.numel() for p in model.parameters())
        st = OM.init_state(cfg, seed=seed)
        assert set(st) == set(model.state_dict()), set(st) ^ set(model.state_dict())
        model.load_state_dict(st, strict=True)
        n1 = n_pts * 2 // 5
        coord = torch.from_numpy(np.concatenate([synth.room_cloud(n1, seed=seed), synth.room_cloud(n_pts - n1, seed=seed + 100)]))
        offset = torch.tensor([n1, n_pts], dtype=torch.int32)
        gen = torch.Generator().manual_seed(seed)
        extra = torch.rand(n_pts, cfg["in_channels"] - 3, generator=gen) * 2 - 1
        feat = torch.cat([coord, extra], 1).contiguous()
        label = torch.randint(0, cfg["num_classes"], (n_pts,), generator=gen)
        label[torch.rand(n_pts, generator=gen) < 0.1] = -1
        out = dict(coord=coord, offset=offset, feat=feat, label=label, state_seed=seed, digest=state_digest(st))
        watch = ["patch_embed.proj.0.weight", "enc_stages.0.down.fc.weight", "seg_head.3.weight", "seg_head.3.bias",
                 "enc_stages.1.blocks.blocks.0.attn.linear_p_bias.3.weight",
                 "dec_stages.0.blocks.blocks.0.attn.weight_encoding.0.weight",
                 "dec_stages.1.up.proj.0.weight", "patch_embed.blocks.blocks.0.attn.linear_q.0.weight"]
        for mode in ("train", "eval"):
            model.load_state_dict(st, strict=True)
            model.train(mode == "train")
            logits = model(dict(coord=coord, feat=feat, offset=offset))
            loss = torch.nn.functional.cross_entropy(logits, label, ignore_index=-1)
            out["logits_" + mode] = logits
            out["loss_" + mode] = loss
            if mode == "train":
                params = dict(model.named_parameters())
                grads = torch.autograd.grad(loss, [params[w] for w in watch], retain_graph=True)
                for w, gr in zip(watch, grads):
                    out["grad_" + w] = gr
                # one optimizer step with the reference recipe (configs/s3dis/semseg-pt-v2m2-0-base.py:41-42:
                # AdamW lr 0.006, weight_decay 0.05) and the loss of the next forward: SURVEY 8f-1
                opt = torch.optim.AdamW(model.parameters(), lr=0.006, weight_decay=0.05)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                for w in watch:
                    out["step1_" + w] = params[w].detach().clone()
                logits2 = model(dict(coord=coord, feat=feat, offset=offset))
                out["loss_step2"] = torch.nn.functional.cross_entropy(logits2, label, ignore_index=-1)
        save("ptv2_%s.npz" % tag, **out)
    with open(os.path.join(HERE, "state_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print("manifest keys:", {k: (len(v) if isinstance(v, dict) else v) for k, v in manifest.items()})


def load_reference_segmentors():
    """pointcept/models/default.py with the reference's REAL registry / builders (pointcept/utils/registry.py,
    models/builder.py, models/losses/builder.py, models/losses/misc.py), loaded file by file because
    `import pointcept.models` pulls spconv.  Replaces the no-op registry stub, then re-imports the PT-v2m2 module so
    that it registers itself as "PT-v2m2"."""
    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    utils = types.ModuleType("pointcept.utils")
    utils.__path__ = [os.path.join(REF, "pointcept/utils")]
    sys.modules["pointcept.utils"] = utils
    load("pointcept.utils.misc", "pointcept/utils/misc.py")
    load("pointcept.utils.registry", "pointcept/utils/registry.py")
    load("pointcept.utils.ply", "pointcept/utils/ply.py")
    load("pointcept.models.builder", "pointcept/models/builder.py")
    sys.modules["pointcept.models.losses"] = types.ModuleType("pointcept.models.losses")
    lb = load("pointcept.models.losses.builder", "pointcept/models/losses/builder.py")
    load("pointcept.models.losses.misc", "pointcept/models/losses/misc.py")
    sys.modules["pointcept.models.losses"].build_criteria = lb.build_criteria
    load_reference_model_module()
    return load("pointcept.models.default", "pointcept/models/default.py")


def gen_segmentor():
    """DefaultSegmentorSAM_Image (pointcept/models/default.py:15-76), built through the reference's registry from the
    model dict of configs/s3dis/semseg-pt-v2m2-0-sam-final.py:10-38, on the inputs of ptv2_s3dis.npz plus the REAL
    batch keys (scene_id, instance); then the trainer's basket statement (engines/train_sam_real.py:229-234) on a
    -100-filled basket.  Also DefaultSegmentor's three return forms."""
    default = load_reference_segmentors()
    g = np.load(os.path.join(HERE, "ptv2_s3dis.npz"))
    cfg = dict(OM.S3DIS_CFG, drop_path_rate=0.0)
    criteria = [dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1)]
    seg = default.MODELS.build(dict(type="DefaultSegmentorSAM_Image", backbone=dict(type="PT-v2m2", **cfg), criteria=criteria))
    st = OM.init_state(cfg, seed=int(g["state_seed"]))
    seg.backbone.load_state_dict(st, strict=True)
    assert all(k.startswith("backbone.") for k in seg.state_dict())
    coord, feat, offset, label = (torch.from_numpy(g[k]) for k in ("coord", "feat", "offset", "label"))
    n = coord.shape[0]
    gen = torch.Generator().manual_seed(77)
    scene_id = ["Area_1/office_3.pth", "Area_4/hallway_11.pth"]
    scene_points = [9000, 7000]  # points of the un-cropped scenes
    bounds = [0] + offset.tolist()
    instance = torch.cat([torch.randperm(scene_points[i], generator=gen)[: bounds[i + 1] - bounds[i]] for i in range(2)])
    batch = dict(coord=coord, feat=feat, offset=offset, segment=label, scene_id=scene_id, instance=instance)
    seg.train()
    out, seg_dict = seg(batch)
    keys = list(seg_dict)
    basket = {k: np.full((m, cfg["num_classes"]), -100.0, np.float32) for k, m in zip(keys, scene_points)}
    for k, v in seg_dict.items():  # train_sam_real.py:231-234
        s_, ori_idx = v[0], v[1]
        basket[k][ori_idx.cpu().detach().numpy()] = s_.cpu().detach().numpy()
    fix = dict(scene_id=np.array(scene_id), scene_points=np.array(scene_points), instance=instance, keys=np.array(keys),
               loss_train=out["loss"], state_seed=int(g["state_seed"]))
    for i, k in enumerate(keys):
        fix["seg_logits_%d" % i], fix["seg_ids_%d" % i], fix["basket_%d" % i] = seg_dict[k][0], seg_dict[k][1], basket[k]
    seg.backbone.load_state_dict(st, strict=True)
    seg.eval()
    with torch.no_grad():
        ev = seg(batch)
        te = seg(dict(coord=coord, feat=feat, offset=offset))
    assert sorted(ev) == ["loss", "seg_logits"] and sorted(te) == ["seg_logits"]
    fix.update(loss_eval=ev["loss"], eval_keys=np.array(sorted(ev)), test_keys=np.array(sorted(te)))
    # loss_weight is honoured (losses/misc.py:38)
    seg2 = default.MODELS.build(dict(type="DefaultSegmentor", backbone=dict(type="PT-v2m2", **cfg),
                                     criteria=[dict(type="CrossEntropyLoss", loss_weight=0.5, ignore_index=-1)]))
    seg2.backbone.load_state_dict(st, strict=True)
    seg2.eval()
    with torch.no_grad():
        fix["loss_eval_half_weight"] = seg2(batch)["loss"]
    save("segmentor_sam.npz", **fix)


def main():
    assert os.path.isdir(REF), "run in the build container: needs /root/reference"
    OP.build()
    install_stubs()
    pointops = sys.modules["pointops"]  # the reference's own python package
    assert pointops.__file__.startswith(REF)
    if os.environ.get("AO_GOLDEN_ONLY") == "gva_pe":  # (adds the round-5 fixture without rewriting the others)
        gen_gva_pe(load_reference_model_module())
        return
    gen_ops(pointops)
    ref = load_reference_model_module()
    gen_gva(ref)
    gen_gva_pe(ref)
    gen_model(ref)
    gen_segmentor()


if __name__ == "__main__":
    main()
