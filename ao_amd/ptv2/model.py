"""PT-v2m2 backbone on the MI355X ops.

Same registry name ("PT-v2m2"), constructor kwargs, `forward(data_dict) -> seg_logits` contract and
state_dict key layout as the reference
(pointcept/models/point_transformer_v2/point_transformer_v2m2_base.py:447-576), so reference
checkpoints load unchanged and the AO trainers can build it from their config.  What differs is
the execution plan:

  * all geometry (kNN tables, grid-pool clusters, interpolation indices/weights) is computed once
    per batch up front (ao_amd/ptv2/geometry.py) instead of inside each layer;
  * grouped vector attention runs through `ao_amd.ptv2.gva` (fused HIP kernels) -- the unfused
    composition of gather ops below (`gva_unfused`) is kept as the in-framework statement of
    the same math for debugging (AO_AMD_GVA=unfused);
  * dense per-point Linear layers run on the fp32-MFMA row GEMM of ao_amd/csrc/gemm.hip (`RowLinear`), with the
    BatchNorm around them fused into its epilogue / operand load (ao_amd/csrc/dense.hip).

Every stage module also answers the reference's own `forward` signature -- `GridPool(points, start=None) ->
(points, cluster)`, `UnpoolWithSkip(points, skip_points, cluster=None)`, `Encoder(points) -> (points, cluster)`,
`Decoder(points, skip_points, cluster)`, `GVAPatchEmbed(points)` (reference :244,305,356,400,441) -- as adapters over
the same kernels, so code that drives a stage directly keeps working; `PointTransformerV2.forward` itself goes
through the geometry plan (`pool` / `unpool` methods) so that nothing geometric is recomputed.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, pointops
from ..pointops.interpolation import _InterpolateRows, interpolation_index_weight
from .geometry import Level, build_geometry, grid_pool_geometry, grid_pool_geometry_torch
from .layers import RowBatchNorm1d, RowLinear, bn_residual_relu, lin_bn_relu


class PointBatchNorm(nn.Module):
    """BatchNorm over the channel (last) axis of (N,C) or (N,L,C); statistics over all other axes
    (reference :26-45 reaches the same statistics through two transposes)."""

    def __init__(self, embed_channels):
        super().__init__()
        self.norm = RowBatchNorm1d(embed_channels)

    def forward(self, input, relu=False):
        if input.dim() == 2:
            return self.norm(input, relu)
        if input.dim() == 3:
            n, l, c = input.shape
            return self.norm(input.reshape(n * l, c), relu).view(n, l, c)
        raise NotImplementedError


class LinBnRelu(nn.Sequential):
    """Sequential(Linear, PointBatchNorm, ReLU) -- the reference's layout and key names ("0.weight",
    "1.norm.*") -- executed as Linear + one fused BatchNorm/ReLU pass."""

    def __init__(self, cin, cout, bias):
        super().__init__(RowLinear(cin, cout, bias=bias), PointBatchNorm(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        if x.dim() == 2:
            return lin_bn_relu(self[0], self[1].norm, x)
        return self[1](self[0](x), relu=True)


def _lin_bn_relu(cin, cout, bias):
    return LinBnRelu(cin, cout, bias)


class GroupedVectorAttention(nn.Module):
    def __init__(self, embed_channels, groups, attn_drop_rate=0.0, qkv_bias=True, pe_multiplier=False,
                 pe_bias=True):
        super().__init__()
        assert embed_channels % groups == 0
        self.embed_channels, self.groups = embed_channels, groups
        self.attn_drop_rate, self.qkv_bias = attn_drop_rate, qkv_bias
        self.pe_multiplier, self.pe_bias = pe_multiplier, pe_bias
        c, g = embed_channels, groups
        self.linear_q = _lin_bn_relu(c, c, qkv_bias)
        self.linear_k = _lin_bn_relu(c, c, qkv_bias)
        self.linear_v = RowLinear(c, c, bias=qkv_bias)
        if pe_multiplier:
            self.linear_p_multiplier = nn.Sequential(nn.Linear(3, c), PointBatchNorm(c), nn.ReLU(inplace=True),
                                                     nn.Linear(c, c))
        if pe_bias:
            self.linear_p_bias = nn.Sequential(nn.Linear(3, c), PointBatchNorm(c), nn.ReLU(inplace=True),
                                               nn.Linear(c, c))
        self.weight_encoding = nn.Sequential(nn.Linear(c, g), PointBatchNorm(g), nn.ReLU(inplace=True),
                                             nn.Linear(g, g))
        self.softmax = nn.Softmax(dim=1)
        self.attn_drop = nn.Dropout(attn_drop_rate)

    def gva_unfused(self, query, key, value, coord, reference_index):
        """The reference's op sequence (:109-128) on the HIP gather kernel."""
        key = pointops.grouping(reference_index, key, coord, with_xyz=True)
        value = pointops.grouping(reference_index, value, coord, with_xyz=False)
        pos, key = key[:, :, 0:3], key[:, :, 3:]
        relation_qk = key - query.unsqueeze(1)
        if self.pe_multiplier:
            relation_qk = relation_qk * self.linear_p_multiplier(pos)
        if self.pe_bias:
            peb = self.linear_p_bias(pos)
            relation_qk = relation_qk + peb
            value = value + peb
        weight = self.weight_encoding(relation_qk)
        weight = self.softmax(weight)
        seed = self.__dict__.get("_ao_drop_seed")  # (tests: the fused kernels' mask instead of nn.Dropout's generator)
        if seed is not None and self.training and self.attn_drop_rate > 0.0:
            from . import gva

            weight = weight * gva.attn_drop_mask(seed, weight.shape[0], weight.shape[1], weight.shape[2], self.attn_drop_rate,
                                                 weight.device)
        else:
            weight = self.attn_drop(weight)
        mask = torch.sign(reference_index + 1).to(weight.dtype)
        weight = weight * mask.unsqueeze(-1)
        n, ns, c = value.shape
        value = value.view(n, ns, self.groups, c // self.groups)
        return (value * weight.unsqueeze(-1)).sum(1).reshape(n, c)

    def forward(self, feat, coord, reference_index):
        query, key, value = self.linear_q(feat), self.linear_k(feat), self.linear_v(feat)
        mode = os.environ.get("AO_AMD_GVA", "fused")
        # attention dropout: inside the fused kernels (mode "fused": one native call per direction); the staged composition
        # (one autograd node per stage) has no place for the mask and falls back to the literal op sequence
        dropping = self.attn_drop_rate > 0.0 and self.training
        fusable = self.pe_bias and not self.pe_multiplier and (not dropping or mode == "fused")
        if mode in ("fused", "staged") and fusable:
            from . import gva

            if gva.supported(self.embed_channels, self.groups, reference_index.shape[1]) and (
                    not dropping or gva.dropout_supported(self.embed_channels, self.groups, reference_index.shape[1])):
                return gva.grouped_vector_attention(self, query, key, value, coord, reference_index)
        return self.gva_unfused(query, key, value, coord, reference_index)


class DropPath(nn.Module):
    """Per-row stochastic depth (timm DropPath semantics, reference :19,160-162): each point's residual
    branch is zeroed with probability p and the survivors are scaled by 1/(1-p)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = torch.empty((x.shape[0],) + (1,) * (x.dim() - 1), device=x.device, dtype=x.dtype).bernoulli_(keep)
        return x * (mask / keep)


class Block(nn.Module):
    def __init__(self, embed_channels, groups, qkv_bias=True, pe_multiplier=False, pe_bias=True,
                 attn_drop_rate=0.0, drop_path_rate=0.0, enable_checkpoint=False):
        super().__init__()
        self.attn = GroupedVectorAttention(embed_channels, groups, attn_drop_rate, qkv_bias, pe_multiplier, pe_bias)
        self.fc1 = RowLinear(embed_channels, embed_channels, bias=False)
        self.fc3 = RowLinear(embed_channels, embed_channels, bias=False)
        self.norm1 = PointBatchNorm(embed_channels)
        self.norm2 = PointBatchNorm(embed_channels)
        self.norm3 = PointBatchNorm(embed_channels)
        self.act = nn.ReLU(inplace=True)
        self.enable_checkpoint = enable_checkpoint
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0.0 else nn.Identity()

    def forward(self, points, reference_index, rowscale=None):
        """rowscale: optional precomputed per-point DropPath factor (BlockSequence draws the factors of all its
        blocks in one go); drawn here when absent."""
        coord, feat, offset = points
        identity = feat
        if rowscale is None and self.training and isinstance(self.drop_path, DropPath) and self.drop_path.drop_prob > 0.0:
            keep = 1.0 - self.drop_path.drop_prob  # timm DropPath: per-point Bernoulli(keep) / keep
            rowscale = torch.empty(feat.shape[0], device=feat.device, dtype=torch.float32).bernoulli_(keep).div_(keep)
        if os.environ.get("AO_AMD_BLOCK", "native") == "native" and os.environ.get("AO_AMD_GVA", "fused") == "fused":
            from . import block as native

            if native.supported(self, feat, reference_index):  # whole block behind one native call per direction
                return [coord, native.block_forward(self, feat, coord, reference_index, rowscale), offset]
        feat = self.norm1(self.fc1(feat), relu=True)
        if self.enable_checkpoint and self.training:
            feat = torch.utils.checkpoint.checkpoint(self.attn, feat, coord, reference_index, use_reentrant=False)
        else:
            feat = self.attn(feat, coord, reference_index)
        feat = self.norm2(feat, relu=True)
        feat = bn_residual_relu(self.norm3.norm, self.fc3(feat), identity, rowscale)
        return [coord, feat, offset]


class BlockSequence(nn.Module):
    def __init__(self, depth, embed_channels, groups, neighbours=16, qkv_bias=True, pe_multiplier=False,
                 pe_bias=True, attn_drop_rate=0.0, drop_path_rate=0.0, enable_checkpoint=False):
        super().__init__()
        if isinstance(drop_path_rate, (list, tuple)):
            rates = list(drop_path_rate)
            assert len(rates) == depth
        elif isinstance(drop_path_rate, float):
            rates = [drop_path_rate] * depth
        else:
            rates = [0.0] * depth
        self.neighbours = neighbours
        self.blocks = nn.ModuleList(
            Block(embed_channels, groups, qkv_bias, pe_multiplier, pe_bias, attn_drop_rate, rates[i], enable_checkpoint)
            for i in range(depth))

    def forward(self, points, reference_index=None):
        coord, feat, offset = points
        if reference_index is None:  # stand-alone use: same call as the reference (:223)
            reference_index, _ = pointops.knn_query(self.neighbours, coord, offset)
        # per-point DropPath factors of all blocks in two launches: Bernoulli(keep_i) / keep_i, row i for block i
        scales = None
        rates = [b.drop_path.drop_prob if isinstance(b.drop_path, DropPath) else 0.0 for b in self.blocks]
        if self.training and any(r > 0.0 for r in rates):
            keep = self.__dict__.get("_keep")
            if keep is None or keep.device != feat.device:
                keep = torch.tensor([1.0 - r for r in rates], dtype=torch.float32, device=feat.device).unsqueeze(1)
                self.__dict__["_keep"] = keep  # plain attribute: not a buffer, so the state_dict stays the reference's
            probs = keep.expand(len(rates), feat.shape[0])
            scales = torch.bernoulli(probs).div_(probs)
        for i, block in enumerate(self.blocks):
            points = block(points, reference_index, scales[i] if scales is not None and rates[i] > 0.0 else None)
        return points


class _SegmentMax(torch.autograd.Function):
    """Per-cluster channel max over the CSR segments of `order` (torch_scatter.segment_csr(reduce="max"),
    reference :266) on ao_amd/csrc/pool.hip; the gradient goes to the arg-max row."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, feat, order, idx_ptr):
        _lib.require_cuda(feat, order, idx_ptr)
        feat = feat.contiguous()
        n_out, c = idx_ptr.shape[0] - 1, feat.shape[1]
        out = torch.empty((n_out, c), dtype=torch.float32, device=feat.device)
        arg = torch.empty((n_out, c), dtype=torch.int32, device=feat.device)
        rc = _lib.lib().pool_max_forward_hip_launcher(n_out, c, feat.data_ptr(), order.data_ptr(), idx_ptr.data_ptr(),
                                                      out.data_ptr(), arg.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "pool_max_forward_hip_launcher")
        ctx.n_in = feat.shape[0]
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        (arg,) = ctx.saved_tensors
        grad = grad.contiguous()
        n_out, c = grad.shape
        gfeat = torch.zeros((ctx.n_in, c), dtype=torch.float32, device=grad.device)
        rc = _lib.lib().pool_max_backward_hip_launcher(n_out, c, grad.data_ptr(), arg.data_ptr(), gfeat.data_ptr(),
                                                       _lib.stream_ptr())
        _lib.check(rc, "pool_max_backward_hip_launcher")
        return gfeat, None, None


class _MapUnpool(torch.autograd.Function):
    """feat[cluster] (the "map" unpool, reference :305-310); backward = per-cluster sum of the fine gradients in the
    pooling's CSR order (ao_amd/csrc/pool.hip: segment_sum) instead of index_put with float atomics."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, feat, cluster, order, idx_ptr):
        ctx.save_for_backward(order, idx_ptr)
        ctx.n_coarse = feat.shape[0]
        return feat[cluster]

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        order, idx_ptr = ctx.saved_tensors
        grad = grad.contiguous()
        out = torch.empty((ctx.n_coarse, grad.shape[1]), dtype=torch.float32, device=grad.device)
        rc = _lib.lib().segment_sum_hip_launcher(ctx.n_coarse, grad.shape[1], grad.data_ptr(), order.data_ptr(),
                                                 idx_ptr.data_ptr(), out.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "segment_sum_hip_launcher")
        return out, None, None, None


class GridPool(nn.Module):
    """Partition-based pooling (reference :229-269).  The clustering itself lives in geometry.py."""

    def __init__(self, in_channels, out_channels, grid_size, bias=False):
        super().__init__()
        self.in_channels, self.out_channels, self.grid_size = in_channels, out_channels, grid_size
        self.fc = RowLinear(in_channels, out_channels, bias=bias)
        self.norm = PointBatchNorm(out_channels)
        self.act = nn.ReLU(inplace=True)

    def pool(self, feat, fine_level):
        """Plan path: the clustering of `fine_level` (geometry.py) is already known."""
        feat = lin_bn_relu(self.fc, self.norm.norm, feat)
        return _SegmentMax.apply(feat, fine_level.order32, fine_level.idx_ptr32)

    def forward(self, points, start=None):
        """Reference signature (:244-269): points = [coord, feat, offset] -> ([coord', feat', offset'], cluster).
        `start` (per-cloud origin of the voxel grid, (B,3)) defaults to the per-cloud minimum, as there."""
        coord, feat, offset = points
        if start is None:
            new_coord, new_offset, cluster, order, idx_ptr = grid_pool_geometry(coord.contiguous(), offset, self.grid_size)
        else:
            new_coord, new_offset, cluster, order, idx_ptr = grid_pool_geometry_torch(coord, offset, self.grid_size, start)
        level = Level(coord=coord, offset=offset, cluster=cluster, order32=order, idx_ptr32=idx_ptr)
        return [new_coord, self.pool(feat, level), new_offset], cluster


class UnpoolWithSkip(nn.Module):
    """Map / interpolation unpooling with skip connection (reference :272-316)."""

    def __init__(self, in_channels, skip_channels, out_channels, bias=True, skip=True, backend="map"):
        super().__init__()
        assert backend in ("map", "interp")
        self.in_channels, self.skip_channels, self.out_channels = in_channels, skip_channels, out_channels
        self.skip, self.backend = skip, backend
        self.proj = _lin_bn_relu(in_channels, out_channels, bias)
        self.proj_skip = _lin_bn_relu(skip_channels, out_channels, bias)

    def forward(self, points, skip_points, cluster=None):
        """Reference signature (:305-316): returns [skip_coord, feat, skip_offset]."""
        coord, feat, offset = points
        skip_coord, skip_feat, skip_offset = skip_points
        level = Level(coord=skip_coord, offset=skip_offset)
        if self.backend == "map" and cluster is not None:
            # CSR of the cluster map, so that the backward is the ordered per-cluster sum of the plan path
            level.cluster = cluster
            level.order32 = torch.sort(cluster, stable=True)[1].int()
            counts = torch.bincount(cluster, minlength=feat.shape[0])
            level.idx_ptr32 = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).int()
        else:
            with torch.no_grad():
                level.up_idx, level.up_weight = interpolation_index_weight(coord, skip_coord, offset.int(), skip_offset.int(), 3)
                from . import gva

                gva.inverse_table(level.up_idx)  # backward gathers in a fixed order (no float atomics)
        return [skip_coord, self.unpool(feat, skip_feat, level), skip_offset]

    def unpool(self, feat, skip_feat, fine_level):
        """Plan path: cluster map / 3-NN table of `fine_level` (geometry.py) already known."""
        feat = self.proj(feat)
        if self.backend == "map" and fine_level.cluster is not None:
            if feat.is_cuda and feat.dtype == torch.float32 and fine_level.order32 is not None:
                feat = _MapUnpool.apply(feat, fine_level.cluster, fine_level.order32, fine_level.idx_ptr32)
            else:
                feat = feat[fine_level.cluster]
        else:
            feat = _InterpolateRows.apply(feat, fine_level.up_idx, fine_level.up_weight)
        if self.skip:
            feat = feat + self.proj_skip(skip_feat)
        return feat


class Encoder(nn.Module):
    def __init__(self, depth, in_channels, embed_channels, groups, grid_size=None, neighbours=16, qkv_bias=True,
                 pe_multiplier=False, pe_bias=True, attn_drop_rate=None, drop_path_rate=None, enable_checkpoint=False):
        super().__init__()
        self.down = GridPool(in_channels, embed_channels, grid_size)
        self.blocks = BlockSequence(depth, embed_channels, groups, neighbours, qkv_bias, pe_multiplier, pe_bias,
                                    attn_drop_rate if attn_drop_rate is not None else 0.0,
                                    drop_path_rate if drop_path_rate is not None else 0.0, enable_checkpoint)

    def forward(self, points):
        """Reference signature (:356-358)."""
        points, cluster = self.down(points)
        return self.blocks(points), cluster


class Decoder(nn.Module):
    def __init__(self, in_channels, skip_channels, embed_channels, groups, depth, neighbours=16, qkv_bias=True,
                 pe_multiplier=False, pe_bias=True, attn_drop_rate=None, drop_path_rate=None, enable_checkpoint=False,
                 unpool_backend="map"):
        super().__init__()
        self.up = UnpoolWithSkip(in_channels, skip_channels, embed_channels, backend=unpool_backend)
        self.blocks = BlockSequence(depth, embed_channels, groups, neighbours, qkv_bias, pe_multiplier, pe_bias,
                                    attn_drop_rate if attn_drop_rate is not None else 0.0,
                                    drop_path_rate if drop_path_rate is not None else 0.0, enable_checkpoint)

    def forward(self, points, skip_points, cluster):
        """Reference signature (:400-402)."""
        return self.blocks(self.up(points, skip_points, cluster))


class GVAPatchEmbed(nn.Module):
    def __init__(self, depth, in_channels, embed_channels, groups, neighbours=16, qkv_bias=True, pe_multiplier=False,
                 pe_bias=True, attn_drop_rate=0.0, drop_path_rate=0.0, enable_checkpoint=False):
        super().__init__()
        self.in_channels, self.embed_channels = in_channels, embed_channels
        self.proj = _lin_bn_relu(in_channels, embed_channels, False)
        self.blocks = BlockSequence(depth, embed_channels, groups, neighbours, qkv_bias, pe_multiplier, pe_bias,
                                    attn_drop_rate, drop_path_rate, enable_checkpoint)

    def forward(self, points):
        """Reference signature (:441-444)."""
        coord, feat, offset = points
        return self.blocks([coord, self.proj(feat), offset])


class PointTransformerV2(nn.Module):
    """Registry name "PT-v2m2" (reference :447)."""

    def __init__(self, in_channels, num_classes, patch_embed_depth=1, patch_embed_channels=48, patch_embed_groups=6,
                 patch_embed_neighbours=8, enc_depths=(2, 2, 6, 2), enc_channels=(96, 192, 384, 512),
                 enc_groups=(12, 24, 48, 64), enc_neighbours=(16, 16, 16, 16), dec_depths=(1, 1, 1, 1),
                 dec_channels=(48, 96, 192, 384), dec_groups=(6, 12, 24, 48), dec_neighbours=(16, 16, 16, 16),
                 grid_sizes=(0.06, 0.12, 0.24, 0.48), attn_qkv_bias=True, pe_multiplier=False, pe_bias=True,
                 attn_drop_rate=0.0, drop_path_rate=0, enable_checkpoint=False, unpool_backend="map", native_param_grads=None):
        """The reference's keyword arguments (:448-474) and ONE more, optional: native_param_grads = "autograd" (default: the
        parameters receive their gradients through AccumulateGrad, as from any module) | "direct" (the native backward
        assigns `p.grad` itself: 840 AccumulateGrad nodes, ~1.4 ms of host time per step, are not built; parameter hooks do
        not fire and torch.autograd.grad(loss, parameters) is not available) -- a config key for loops that only call
        loss.backward(): backbone = dict(type="PT-v2m2", ..., native_param_grads="direct")."""
        super().__init__()
        if native_param_grads is not None:
            assert native_param_grads in ("autograd", "direct"), native_param_grads
            self.native_param_grads = native_param_grads
        self.in_channels, self.num_classes = in_channels, num_classes
        self.num_stages = len(enc_depths)
        for seq in (dec_depths, enc_channels, dec_channels, enc_groups, dec_groups, enc_neighbours, dec_neighbours,
                    grid_sizes):
            assert len(seq) == self.num_stages
        self.grid_sizes, self.unpool_backend = tuple(grid_sizes), unpool_backend
        self.patch_embed = GVAPatchEmbed(patch_embed_depth, in_channels, patch_embed_channels, patch_embed_groups,
                                         patch_embed_neighbours, attn_qkv_bias, pe_multiplier, pe_bias, attn_drop_rate,
                                         enable_checkpoint=enable_checkpoint)
        enc_dp = [x.item() for x in torch.linspace(0, drop_path_rate, sum(enc_depths))]
        dec_dp = [x.item() for x in torch.linspace(0, drop_path_rate, sum(dec_depths))]
        enc_channels = [patch_embed_channels] + list(enc_channels)
        dec_channels = list(dec_channels) + [enc_channels[-1]]
        self.enc_stages, self.dec_stages = nn.ModuleList(), nn.ModuleList()
        for i in range(self.num_stages):
            self.enc_stages.append(Encoder(
                enc_depths[i], enc_channels[i], enc_channels[i + 1], enc_groups[i], grid_sizes[i], enc_neighbours[i],
                attn_qkv_bias, pe_multiplier, pe_bias, attn_drop_rate,
                enc_dp[sum(enc_depths[:i]):sum(enc_depths[:i + 1])], enable_checkpoint))
            self.dec_stages.append(Decoder(
                dec_channels[i + 1], enc_channels[i], dec_channels[i], dec_groups[i], dec_depths[i], dec_neighbours[i],
                attn_qkv_bias, pe_multiplier, pe_bias, attn_drop_rate,
                dec_dp[sum(dec_depths[:i]):sum(dec_depths[:i + 1])], enable_checkpoint, unpool_backend))
        self.seg_head = (nn.Sequential(RowLinear(dec_channels[0], dec_channels[0]), PointBatchNorm(dec_channels[0]),
                                       nn.ReLU(inplace=True), RowLinear(dec_channels[0], num_classes))
                         if num_classes > 0 else nn.Identity())

    def geometry_neighbours(self):
        """K values needed per level (level 0: patch embedding + last decoder stage; level i + 1: encoder stage i + decoder
        stage i + 1)."""
        ks = [{self.patch_embed.blocks.neighbours, self.dec_stages[0].blocks.neighbours}]
        for i in range(self.num_stages):
            here = {self.enc_stages[i].blocks.neighbours}
            if i + 1 < self.num_stages:
                here.add(self.dec_stages[i + 1].blocks.neighbours)
            ks.append(here)
        return [sorted(k) for k in ks]

    def geometry(self, coord, offset):
        return build_geometry(coord, offset, self.grid_sizes, self.geometry_neighbours(), interp=self.unpool_backend == "interp")

    # ---- DistributedDataParallel (what the reference's create_ddp_model wraps every rank's model in, engines/defaults.py:20-43)
    @property
    def _ddp_params_and_buffers_to_ignore(self):
        """DistributedDataParallel reads this attribute of the module it wraps (torch/nn/parallel/distributed.py) and leaves
        the named parameters alone: no per-parameter hook, no bucket copy, no all-reduce of its own.  The native backward
        writes all 840 gradients into ONE flat buffer, so it averages that buffer over the ranks with one RCCL all-reduce
        itself (native_model._NativeModel.backward) -- DDP's 840 AccumulateGrad hooks and bucket copies cost 3.9 ms of a 10.7 ms
        step on one MI355X (bench.py `reference_loop.ddp`).  The trainer does not change: it still wraps the model, and the
        wrapper still does its start-up checks on the one parameter left to it (DDP refuses a module without any).
        AO_AMD_DDP_SYNC=ddp: hand every parameter to DDP as any other module does."""
        return parallel_ddp_ignore(self, "")

    def forward(self, data_dict, geometry=None):
        coord, feat = data_dict["coord"], data_dict["feat"]
        offset = data_dict["offset"].int()
        from . import native_model

        if geometry is None:
            geometry = data_dict.get("geometry")  # prebuilt SceneGeometry (parallel.GeometryPrefetcher: an optional acceleration)
        if geometry is None and native_model.supported(self, feat) and native_model.pipelined_ok(self):
            # the default: geometry built inside the native forward, its size-dependent half on a side stream behind the
            # level-0 prefix of the network (native_model._NativeModel.forward)
            return native_model.forward(self, data_dict, None)
        geo = geometry if geometry is not None else self.geometry(coord, offset)

        if native_model.supported(self, feat) and native_model.geometry_supported(geo):  # the whole network behind one native call per direction
            return native_model.forward(self, data_dict, geo)
        if self.__dict__.get("_ao_ddp_native_sync") and self.training and torch.is_grad_enabled():
            raise RuntimeError("ao_amd: this batch takes the stage-wise python path, whose gradients the native flat all-reduce "
                               "does not see, under a DistributedDataParallel wrapper that was told to leave the parameters "
                               "alone; set AO_AMD_DDP_SYNC=ddp (DDP then synchronises every parameter itself)")
        lv = geo.levels
        pe = self.patch_embed
        feat = pe.blocks([lv[0].coord, pe.proj(feat), lv[0].offset], lv[0].neighbours(pe.blocks.neighbours))[1]
        skips = [feat]
        for i, enc in enumerate(self.enc_stages):
            feat = enc.down.pool(feat, lv[i])
            feat = enc.blocks([lv[i + 1].coord, feat, lv[i + 1].offset], lv[i + 1].neighbours(enc.blocks.neighbours))[1]
            skips.append(feat)
        feat = skips.pop()
        for i in reversed(range(self.num_stages)):
            dec = self.dec_stages[i]
            feat = dec.up.unpool(feat, skips.pop(), lv[i])
            feat = dec.blocks([lv[i].coord, feat, lv[i].offset], lv[i].neighbours(dec.blocks.neighbours))[1]
        if isinstance(self.seg_head, nn.Sequential):
            return self.seg_head[3](lin_bn_relu(self.seg_head[0], self.seg_head[1].norm, feat))
        return self.seg_head(feat)


def parallel_ddp_ignore(module, prefix):
    """The `_ddp_params_and_buffers_to_ignore` list of `module` (PointTransformerV2 itself, prefix "", or a segmentor that holds
    it as `backbone`, prefix "backbone."): every backbone parameter but the smallest one, when the native runtime will average
    the flat gradient buffer itself -- a process group exists, the network is in the native runtime's shape, AO_AMD_DDP_SYNC
    is not "ddp".  DDP asks for it while it is being constructed, which is also when it broadcasts rank 0's parameters to the
    others: the parameters it is told to ignore are broadcast here instead."""
    import torch.distributed as dist

    from . import native_model

    backbone = module if prefix == "" else module.backbone
    if os.environ.get("AO_AMD_DDP_SYNC", "native") == "ddp" or not (dist.is_available() and dist.is_initialized()):
        return []
    if os.environ.get("AO_AMD_MODEL", "native") != "native" or os.environ.get("AO_AMD_BLOCK", "native") != "native" \
            or os.environ.get("AO_AMD_GVA", "fused") != "fused":
        return []
    named = list(backbone.named_parameters())
    if not named or not all(p.is_cuda and p.dtype == torch.float32 and p.requires_grad for _, p in named):
        return []
    if not native_model.runtime(backbone).static_ok:
        return []
    keep = min(range(len(named)), key=lambda i: (named[i][1].numel(), i))  # DDP needs at least one parameter of its own
    names = [prefix + n for i, (n, _) in enumerate(named) if i != keep]
    if not backbone.__dict__.get("_ao_ddp_native_sync"):
        backbone.__dict__["_ao_ddp_native_sync"] = True
        if dist.get_world_size() > 1:  # what DDP's _sync_module_states does for the parameters it owns
            with torch.no_grad():
                ps = [p.data for i, (_, p) in enumerate(named) if i != keep]
                flat = torch.cat([t.reshape(-1) for t in ps])
                dist.broadcast(flat, 0)
                torch._foreach_copy_(ps, [c.view_as(t) for c, t in zip(flat.split([t.numel() for t in ps]), ps)])
    return names


MODEL_TYPE = "PT-v2m2"


def build_from_cfg(cfg):
    """cfg: the `backbone=dict(type="PT-v2m2", ...)` dict of the reference configs."""
    cfg = dict(cfg)
    assert cfg.pop("type", MODEL_TYPE) == MODEL_TYPE
    return PointTransformerV2(**cfg)
