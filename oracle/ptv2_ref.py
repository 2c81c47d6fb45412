"""oracle/ptv2_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional torch-CPU restatement of the reference's PT-v2m2 backbone
(pointcept/models/point_transformer_v2/point_transformer_v2m2_base.py:26-576).
It is written as plain functions over a flat ``state`` dict that uses the
reference's state_dict key names, so the same weights can be pushed through the
reference nn.Module (in the build container, tests/golden/make_golden.py), this
oracle, and the HIP model (ao_amd/ptv2).  fp32 everywhere.

Pinned by: tests/golden/ptv2_*.npz, produced by running the *reference module
itself* on CPU with dependency stubs (see make_golden.py).  Third-party pieces
that are not vendored in the reference are restated from their documented
behaviour and are "parity unpinned":
  * torch_geometric.nn.pool.voxel_grid -> torch_cluster.grid_cluster
    (call site :257-259): id = sum_d trunc((pos_d-start_d)/size_d) * prod_{e<d} n_e,
    n_e = trunc((end_e-start_e)/size_e)+1, batch index as an extra last dim of size 1;
  * torch_scatter.segment_csr (min / mean / max) (:249-253,265-266);
  * timm DropPath (:19,160-162): per-row Bernoulli(keep) / keep.
"""
import math

import torch
import torch.nn.functional as F

from . import pointops_ref as P

S3DIS_CFG = dict(  # configs/s3dis/semseg-pt-v2m2-0-base.py:12-36
    in_channels=6, num_classes=13, patch_embed_depth=2, patch_embed_channels=48,
    patch_embed_groups=6, patch_embed_neighbours=16, enc_depths=(2, 6, 2),
    enc_channels=(96, 192, 384), enc_groups=(12, 24, 48), enc_neighbours=(16, 16, 16),
    dec_depths=(1, 1, 1), dec_channels=(48, 96, 192), dec_groups=(6, 12, 24),
    dec_neighbours=(16, 16, 16), grid_sizes=(0.1, 0.2, 0.4), attn_qkv_bias=True,
    pe_multiplier=False, pe_bias=True, attn_drop_rate=0.0, drop_path_rate=0.3,
    enable_checkpoint=False, unpool_backend="interp",
)

SCANNET_CFG = dict(  # configs/scannet/semseg-pt-v2m2-0-base.py:10-37
    in_channels=9, num_classes=20, patch_embed_depth=1, patch_embed_channels=48,
    patch_embed_groups=6, patch_embed_neighbours=8, enc_depths=(2, 2, 6, 2),
    enc_channels=(96, 192, 384, 512), enc_groups=(12, 24, 48, 64),
    enc_neighbours=(16, 16, 16, 16), dec_depths=(1, 1, 1, 1),
    dec_channels=(48, 96, 192, 384), dec_groups=(6, 12, 24, 48),
    dec_neighbours=(16, 16, 16, 16), grid_sizes=(0.06, 0.15, 0.375, 0.9375),
    attn_qkv_bias=True, pe_multiplier=False, pe_bias=True, attn_drop_rate=0.0,
    drop_path_rate=0.3, enable_checkpoint=False, unpool_backend="map",
)


# ----------------------------------------------------------- third-party --
def segment_csr(src, indptr, reduce):
    """torch_scatter.segment_csr restated on torch.segment_reduce (sequential per segment)."""
    red = {"mean": "mean", "max": "max", "min": "min", "sum": "sum"}[reduce]
    return torch.segment_reduce(src, red, offsets=indptr.long(), axis=0)


def voxel_grid(pos, size, batch, start=0):
    """torch_geometric voxel_grid(pos, size, batch, start=0) -> torch_cluster.grid_cluster."""
    dim = pos.shape[1]
    p = torch.cat([pos, batch.view(-1, 1).to(pos.dtype)], dim=-1)
    sz = torch.tensor([size] * dim + [1.0], dtype=pos.dtype)
    st = torch.tensor([float(start)] * dim + [0.0], dtype=pos.dtype)
    end = p.max(0)[0]
    p = p - st
    num = ((end - st) / sz).long() + 1
    num = num.cumprod(0)
    num = torch.cat([num.new_ones(1), num])[: sz.numel()]
    out = (p / sz.view(1, -1)).long()
    return (out * num.view(1, -1)).sum(1)


# ----------------------------------------------------------------- layers --
class Ctx:
    """Forward-pass context: weights, mode and (optional) per-row drop-path keep masks."""

    def __init__(self, state, training, drop_masks=None, update_stats=False):
        self.s = state
        self.training = training
        self.drop_masks = drop_masks or {}
        self.update_stats = update_stats


def _linear(cx, x, name):
    return F.linear(x, cx.s[name + ".weight"], cx.s.get(name + ".bias"))


def _pbn(cx, x, name):
    """PointBatchNorm (:26-45): BatchNorm1d on (N,C), or on (N,L,C) through the (N,C,L) transpose
    exactly as the reference does (the transposed layout also selects torch's accurate two-pass
    variance path on CPU; the flat (N*L,C) path loses ~3 digits when |mean| >> std)."""
    pre = name + ".norm"
    rm, rv = cx.s[pre + ".running_mean"], cx.s[pre + ".running_var"]
    if cx.training and not cx.update_stats:
        rm, rv = rm.clone(), rv.clone()
    args = (rm, rv, cx.s[pre + ".weight"], cx.s[pre + ".bias"], cx.training, 0.1, 1e-5)
    if x.dim() == 3:
        return F.batch_norm(x.transpose(1, 2).contiguous(), *args).transpose(1, 2).contiguous()
    return F.batch_norm(x, *args)


def _gva(cx, pre, feat, coord, ref_idx, groups):
    """GroupedVectorAttention.forward (:103-129), pe_bias=True, pe_multiplier=False."""
    q = F.relu(_pbn(cx, _linear(cx, feat, pre + ".linear_q.0"), pre + ".linear_q.1"))
    k = F.relu(_pbn(cx, _linear(cx, feat, pre + ".linear_k.0"), pre + ".linear_k.1"))
    v = _linear(cx, feat, pre + ".linear_v")
    kg = P.grouping(ref_idx, k, coord, with_xyz=True)
    vg = P.grouping(ref_idx, v, coord, with_xyz=False)
    pos, kg = kg[:, :, 0:3], kg[:, :, 3:]
    rel = kg - q.unsqueeze(1)
    peb = _linear(cx, pos, pre + ".linear_p_bias.0")
    peb = F.relu(_pbn(cx, peb, pre + ".linear_p_bias.1"))
    peb = _linear(cx, peb, pre + ".linear_p_bias.3")
    rel = rel + peb
    vg = vg + peb
    w = _linear(cx, rel, pre + ".weight_encoding.0")
    w = F.relu(_pbn(cx, w, pre + ".weight_encoding.1"))
    w = _linear(cx, w, pre + ".weight_encoding.3")
    w = torch.softmax(w, dim=1)
    mask = torch.sign(ref_idx + 1).to(w.dtype)
    w = w * mask.unsqueeze(-1)
    n, ns, c = vg.shape
    vg = vg.view(n, ns, groups, c // groups)
    return (vg * w.unsqueeze(-1)).sum(1).reshape(n, c)


def _block(cx, pre, feat, coord, ref_idx, groups):
    """Block.forward (:164-177)."""
    identity = feat
    feat = F.relu(_pbn(cx, F.linear(feat, cx.s[pre + ".fc1.weight"]), pre + ".norm1"))
    feat = _gva(cx, pre + ".attn", feat, coord, ref_idx, groups)
    feat = F.relu(_pbn(cx, feat, pre + ".norm2"))
    feat = _pbn(cx, F.linear(feat, cx.s[pre + ".fc3.weight"]), pre + ".norm3")
    keep = cx.drop_masks.get(pre)
    if cx.training and keep is not None:  # timm DropPath: x / keep_prob * bernoulli, per row
        feat = feat * keep
    return F.relu(identity + feat)


def _block_sequence(cx, pre, depth, coord, feat, offset, groups, neighbours):
    """BlockSequence.forward (:219-226): one kNN shared by all blocks."""
    ref_idx, _ = P.knn_query(neighbours, coord, offset.int())
    for i in range(depth):
        feat = _block(cx, "%s.blocks.%d" % (pre, i), feat, coord, ref_idx, groups)
    return feat


def grid_pool_geometry(coord, offset, grid_size):
    """The coord-only part of GridPool.forward (:244-269)."""
    batch = P.offset2batch(offset)
    ptr = torch.cat([batch.new_zeros(1), torch.cumsum(batch.bincount(), dim=0)])
    start = segment_csr(coord, ptr, "min")
    cluster = voxel_grid(coord - start[batch], grid_size, batch, start=0)
    unique, cluster, counts = torch.unique(cluster, sorted=True, return_inverse=True, return_counts=True)
    _, order = torch.sort(cluster, stable=True)
    idx_ptr = torch.cat([counts.new_zeros(1), torch.cumsum(counts, dim=0)])
    new_coord = segment_csr(coord[order], idx_ptr, "mean")
    new_batch = batch[order][idx_ptr[:-1]]
    new_offset = P.batch2offset(new_batch)
    return new_coord, new_offset, cluster, order, idx_ptr


def _grid_pool(cx, pre, coord, feat, offset, grid_size):
    feat = F.relu(_pbn(cx, F.linear(feat, cx.s[pre + ".fc.weight"]), pre + ".norm"))
    new_coord, new_offset, cluster, order, idx_ptr = grid_pool_geometry(coord, offset, grid_size)
    new_feat = segment_csr(feat[order], idx_ptr, "max")
    return new_coord, new_feat, new_offset, cluster


def _unpool(cx, pre, coord, feat, offset, skip_coord, skip_feat, skip_offset, cluster, backend):
    """UnpoolWithSkip.forward (:305-316)."""
    proj = F.relu(_pbn(cx, _linear(cx, feat, pre + ".proj.0"), pre + ".proj.1"))
    if backend == "map" and cluster is not None:
        feat = proj[cluster]
    else:
        feat = P.interpolation(coord, skip_coord, proj, offset.int(), skip_offset.int())
    skip = F.relu(_pbn(cx, _linear(cx, skip_feat, pre + ".proj_skip.0"), pre + ".proj_skip.1"))
    return feat + skip


def forward(state, cfg, coord, feat, offset, training=True, drop_masks=None, update_stats=False):
    """PointTransformerV2.forward (:556-576) -> seg_logits (N, num_classes)."""
    cx = Ctx(state, training, drop_masks, update_stats)
    offset = offset.long()
    ns = len(cfg["enc_depths"])
    feat = F.relu(_pbn(cx, F.linear(feat, state["patch_embed.proj.0.weight"]), "patch_embed.proj.1"))
    feat = _block_sequence(cx, "patch_embed.blocks", cfg["patch_embed_depth"], coord, feat, offset,
                           cfg["patch_embed_groups"], cfg["patch_embed_neighbours"])
    skips = [(coord, feat, offset)]
    clusters = []
    for i in range(ns):
        pre = "enc_stages.%d" % i
        coord, feat, offset, cluster = _grid_pool(cx, pre + ".down", coord, feat, offset, cfg["grid_sizes"][i])
        feat = _block_sequence(cx, pre + ".blocks", cfg["enc_depths"][i], coord, feat, offset,
                               cfg["enc_groups"][i], cfg["enc_neighbours"][i])
        clusters.append(cluster)
        skips.append((coord, feat, offset))
    coord, feat, offset = skips.pop()
    for i in reversed(range(ns)):
        pre = "dec_stages.%d" % i
        sc, sf, so = skips.pop()
        feat = _unpool(cx, pre + ".up", coord, feat, offset, sc, sf, so, clusters[i], cfg["unpool_backend"])
        coord, offset = sc, so
        feat = _block_sequence(cx, pre + ".blocks", cfg["dec_depths"][i], coord, feat, offset,
                               cfg["dec_groups"][i], cfg["dec_neighbours"][i])
    if cfg["num_classes"] > 0:
        feat = F.relu(_pbn(cx, _linear(cx, feat, "seg_head.0"), "seg_head.1"))
        feat = _linear(cx, feat, "seg_head.3")
    return feat


# ------------------------------------------------------------- state init --
def block_names(cfg):
    """[(prefix, channels, groups)] for every Block, in forward order of construction."""
    out = []
    for i in range(cfg["patch_embed_depth"]):
        out.append(("patch_embed.blocks.blocks.%d" % i, cfg["patch_embed_channels"], cfg["patch_embed_groups"]))
    for s, d in enumerate(cfg["enc_depths"]):
        for i in range(d):
            out.append(("enc_stages.%d.blocks.blocks.%d" % (s, i), cfg["enc_channels"][s], cfg["enc_groups"][s]))
    for s, d in enumerate(cfg["dec_depths"]):
        for i in range(d):
            out.append(("dec_stages.%d.blocks.blocks.%d" % (s, i), cfg["dec_channels"][s], cfg["dec_groups"][s]))
    return out


def drop_path_rates(cfg):
    """{block prefix: rate}; linspace(0, rate, sum(depths)) per enc / dec (:499-504); patch_embed: 0."""
    rates = {}
    for kind, depths in (("enc", cfg["enc_depths"]), ("dec", cfg["dec_depths"])):
        vals = [x.item() for x in torch.linspace(0, cfg["drop_path_rate"], sum(depths))]
        j = 0
        for s, d in enumerate(depths):
            for i in range(d):
                rates["%s_stages.%d.blocks.blocks.%d" % (kind, s, i)] = vals[j]
                j += 1
    return rates


def _add_linear(st, name, cout, cin, bias, gen):
    bound = 1.0 / math.sqrt(cin)
    st[name + ".weight"] = (torch.rand(cout, cin, generator=gen) * 2 - 1) * bound
    if bias:
        st[name + ".bias"] = (torch.rand(cout, generator=gen) * 2 - 1) * bound


def _add_bn(st, name, c, gen, randomize):
    pre = name + ".norm"
    if randomize:  # non-trivial affine + running stats so eval-mode parity is a real check
        st[pre + ".weight"] = 0.5 + torch.rand(c, generator=gen)
        st[pre + ".bias"] = 0.2 * torch.randn(c, generator=gen)
        st[pre + ".running_mean"] = 0.1 * torch.randn(c, generator=gen)
        st[pre + ".running_var"] = 0.5 + torch.rand(c, generator=gen)
    else:
        st[pre + ".weight"] = torch.ones(c)
        st[pre + ".bias"] = torch.zeros(c)
        st[pre + ".running_mean"] = torch.zeros(c)
        st[pre + ".running_var"] = torch.ones(c)
    st[pre + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def init_state(cfg, seed=0, randomize_bn=True):
    """Random weights under the reference's state_dict names/shapes (constructor :449-554)."""
    gen = torch.Generator().manual_seed(seed)
    st = {}
    c0 = cfg["patch_embed_channels"]
    _add_linear(st, "patch_embed.proj.0", c0, cfg["in_channels"], False, gen)
    _add_bn(st, "patch_embed.proj.1", c0, gen, randomize_bn)
    qb = cfg["attn_qkv_bias"]
    for pre, c, g in block_names(cfg):
        a = pre + ".attn"
        _add_linear(st, a + ".linear_q.0", c, c, qb, gen)
        _add_bn(st, a + ".linear_q.1", c, gen, randomize_bn)
        _add_linear(st, a + ".linear_k.0", c, c, qb, gen)
        _add_bn(st, a + ".linear_k.1", c, gen, randomize_bn)
        _add_linear(st, a + ".linear_v", c, c, qb, gen)
        _add_linear(st, a + ".linear_p_bias.0", c, 3, True, gen)
        _add_bn(st, a + ".linear_p_bias.1", c, gen, randomize_bn)
        _add_linear(st, a + ".linear_p_bias.3", c, c, True, gen)
        _add_linear(st, a + ".weight_encoding.0", g, c, True, gen)
        _add_bn(st, a + ".weight_encoding.1", g, gen, randomize_bn)
        _add_linear(st, a + ".weight_encoding.3", g, g, True, gen)
        _add_linear(st, pre + ".fc1", c, c, False, gen)
        _add_linear(st, pre + ".fc3", c, c, False, gen)
        for nm in ("norm1", "norm2", "norm3"):
            _add_bn(st, pre + "." + nm, c, gen, randomize_bn)
    enc_ch = [c0] + list(cfg["enc_channels"])
    dec_ch = list(cfg["dec_channels"]) + [enc_ch[-1]]
    for i in range(len(cfg["enc_depths"])):
        _add_linear(st, "enc_stages.%d.down.fc" % i, enc_ch[i + 1], enc_ch[i], False, gen)
        _add_bn(st, "enc_stages.%d.down.norm" % i, enc_ch[i + 1], gen, randomize_bn)
        up = "dec_stages.%d.up" % i
        _add_linear(st, up + ".proj.0", dec_ch[i], dec_ch[i + 1], True, gen)
        _add_bn(st, up + ".proj.1", dec_ch[i], gen, randomize_bn)
        _add_linear(st, up + ".proj_skip.0", dec_ch[i], enc_ch[i], True, gen)
        _add_bn(st, up + ".proj_skip.1", dec_ch[i], gen, randomize_bn)
    if cfg["num_classes"] > 0:
        _add_linear(st, "seg_head.0", dec_ch[0], dec_ch[0], True, gen)
        _add_bn(st, "seg_head.1", dec_ch[0], gen, randomize_bn)
        _add_linear(st, "seg_head.3", cfg["num_classes"], dec_ch[0], True, gen)
    return st


def is_param(name):
    return not name.endswith(("running_mean", "running_var", "num_batches_tracked"))


class RefModule(torch.nn.Module):
    """Thin nn.Module shell around `forward` (for optimizers / DDP in tests and the CPU baseline)."""

    def __init__(self, cfg, seed=0, randomize_bn=False):
        super().__init__()
        self.cfg = dict(cfg)
        st = init_state(cfg, seed, randomize_bn)
        self._names = list(st.keys())
        for k, v in st.items():
            key = k.replace(".", "/")
            if is_param(k):
                self.register_parameter(key, torch.nn.Parameter(v))
            else:
                self.register_buffer(key, v)

    def state(self):
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        return {k.replace("/", "."): v for k, v in d.items()}

    def forward(self, data_dict, drop_masks=None):
        return forward(self.state(), self.cfg, data_dict["coord"], data_dict["feat"], data_dict["offset"],
                       training=self.training, drop_masks=drop_masks, update_stats=True)
