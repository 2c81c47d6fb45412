"""PointTransformerV2.forward / backward as ONE native call per direction (ao_amd/csrc/model.hip).

`PointTransformerV2.forward` of model.py dispatches here when the whole network is in the shape the native runtime
covers (fp32 CUDA parameters, every Block supported by the Block runtime, training or eval).  The module tree keeps its
nn.Parameters / buffers under the reference's state_dict names (point_transformer_v2m2_base.py:447-554); this file only
gathers their device pointers into a `ptv2_model` struct (include/ptv2_hip.h), hands the scene geometry over, and wraps
the two launchers in one autograd node.

Parameter gradients are written by the kernels straight into ONE flat fp32 buffer laid out like
`optim.FlatAdamW`'s (parameters in `module.parameters()` order, every slot aligned to 4 floats), so the optimizer
consumes it without a flatten copy.  Two ways to hand them to `.grad`:
  * "autograd" (default): the parameters are inputs of the autograd node and receive views of the flat buffer through
    AccumulateGrad -- hooks, DistributedDataParallel and gradient accumulation work as with any module;
  * "direct" (`backbone.native_param_grads = "direct"`, what bench.py selects for its flat all-reduce): the node assigns
    `p.grad` itself; 840 AccumulateGrad nodes (~1.4 ms of host time per step) are not built.  Parameter hooks do not
    fire in this mode; gradients are accumulated (added) when `p.grad` already holds a value.
"""
import ctypes
import os

import torch

from .. import _lib
from . import block as _block
from . import gva as _gva

_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
MAX_STAGES, MAX_BLOCKS = 5, 40
NPARAM, NBN = _block.NPARAM, _block.NBN


class _LinBn(ctypes.Structure):  # mirrors ptv2_linbn
    _fields_ = [("cin", _I), ("cout", _I)] + [(n, _P) for n in ("w", "b", "gamma", "beta", "run_mean", "run_var", "batches",
                                                                 "gw", "gb", "ggamma", "gbeta")]


class _Level(ctypes.Structure):  # mirrors ptv2_level
    _fields_ = [("n", _I), ("b", _I)] + [(n, _P) for n in ("coord", "offset", "order", "idx_ptr", "cluster", "up_idx", "up_w",
                                                            "up_inv_ptr", "up_inv_rows")]


class _Seq(ctypes.Structure):  # mirrors ptv2_seq
    _fields_ = [(n, _I) for n in ("level", "depth", "first_block", "c", "g", "k")] + [(n, _P) for n in ("idx", "mu", "cov", "inv_ptr",
                                                                                                       "inv_rows")]


class _MBlock(ctypes.Structure):  # mirrors ptv2_model_block
    _fields_ = [("param", _P * NPARAM), ("run_mean", _P * NBN), ("run_var", _P * NBN), ("batches", _P * NBN),
                ("gparam", _P * NPARAM), ("rowscale", _P), ("attn_drop_p", _F), ("attn_drop_seed", ctypes.c_uint)]


class _Model(ctypes.Structure):  # mirrors ptv2_model
    _fields_ = ([(n, _I) for n in ("num_stages", "in_channels", "num_classes", "training", "interp")]
                + [("eps", _F), ("momentum", _F), ("level", _Level * (MAX_STAGES + 1)), ("seq", _Seq * (2 * MAX_STAGES + 1)),
                   ("num_blocks", _I), ("block", _MBlock * MAX_BLOCKS), ("embed", _LinBn), ("down", _LinBn * MAX_STAGES),
                   ("up", _LinBn * MAX_STAGES), ("up_skip", _LinBn * MAX_STAGES), ("head", _LinBn)]
                + [(n, _P) for n in ("head_w", "head_b", "g_head_w", "g_head_b", "feat", "logits", "saved")]
                + [("saved_bytes", ctypes.c_size_t), ("matmul_bf16", _I), ("checkpoint", _I),
                   ("decoder_done_event", _P), ("saved0", _P), ("saved0_bytes", ctypes.c_size_t)])


_lib.register({
    "ptv2_model_saved_bytes": (_lib._c_size, [_P]),
    "ptv2_model_workspace_bytes": (_lib._c_size, [_P]),
    "ptv2_model_forward_hip_launcher": (_lib._c_int, [_P, _P, _lib._c_size, _P]),
    "ptv2_model_backward_hip_launcher": (_lib._c_int, [_P, _P, _P, _lib._c_size, _P]),
    "ptv2_model_prefix_saved_bytes": (_lib._c_size, [_P]),
    "ptv2_model_prefix_workspace_bytes": (_lib._c_size, [_P]),
    "ptv2_model_forward_prefix_hip_launcher": (_lib._c_int, [_P, _P, _lib._c_size, _P]),
    "ptv2_model_forward_rest_hip_launcher": (_lib._c_int, [_P, _P, _lib._c_size, _P]),
})


_lib.check_struct(2, _Model)


def grad_layout(params):
    """Offsets (in floats) of every parameter's slot in the flat gradient buffer: `module.parameters()` order, each
    slot aligned to 4 floats (the kernels store float4) -- the layout of optim.FlatAdamW."""
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + 3) // 4 * 4
    return offsets, off


class _Runtime:
    """Everything that does not change between steps: the struct with the parameter pointers filled in, the list of
    (struct field, gradient slot) pairs, the BlockSequences in ABI order.  Rebuilt when a parameter or buffer moves."""

    def __init__(self, model):
        import weakref

        self.model_ref = weakref.ref(model)  # (the model owns this object: no cycle)
        self.sequences = ([model.patch_embed.blocks] + [e.blocks for e in model.enc_stages]
                          + [d.blocks for d in model.dec_stages])
        S = model.num_stages
        self.S = S
        self.params = list(model.parameters())
        self.offsets, self.total = grad_layout(self.params)
        index = {id(p): i for i, p in enumerate(self.params)}
        M = _Model()
        M.num_stages, M.in_channels, M.num_classes = S, model.in_channels, model.num_classes
        M.interp = 1 if model.unpool_backend == "interp" else 0
        bns = []
        self.grad_fields = []  # (setter, parameter index): where each gradient slot's pointer goes in the struct
        self.tensor_slots = []  # (owner dict, name) of every tensor whose pointer is cached in the struct
        owners = {}
        for m in model.modules():
            for name, t in m._parameters.items():
                if t is not None:
                    owners[id(t)] = (m._parameters, name)

        def ptr(t):
            if t is None:
                return None
            self.tensor_slots.append(owners[id(t)])
            return t.data_ptr()

        def buf(bn, name):
            t = bn._buffers.get(name)
            if t is None:
                return None
            self.tensor_slots.append((bn._buffers, name))
            return t.data_ptr()

        def linbn(dst, linear, bn):
            dst.cout, dst.cin = linear.weight.shape
            dst.w, dst.b, dst.gamma, dst.beta = ptr(linear.weight), ptr(linear.bias), ptr(bn.weight), ptr(bn.bias)
            dst.run_mean, dst.run_var, dst.batches = buf(bn, "running_mean"), buf(bn, "running_var"), buf(bn, "num_batches_tracked")
            for field, p in (("gw", linear.weight), ("gb", linear.bias), ("ggamma", bn.weight), ("gbeta", bn.bias)):
                if p is not None:
                    self.grad_fields.append((dst, field, None, index[id(p)]))
            bns.append(bn)

        linbn(M.embed, model.patch_embed.proj[0], model.patch_embed.proj[1].norm)
        for i in range(S):
            enc, dec = model.enc_stages[i], model.dec_stages[i]
            linbn(M.down[i], enc.down.fc, enc.down.norm.norm)
            linbn(M.up[i], dec.up.proj[0], dec.up.proj[1].norm)
            linbn(M.up_skip[i], dec.up.proj_skip[0], dec.up.proj_skip[1].norm)
        linbn(M.head, model.seg_head[0], model.seg_head[1].norm)
        cls = model.seg_head[3]
        M.head_w, M.head_b = ptr(cls.weight), ptr(cls.bias)
        self.grad_fields.append((M, "g_head_w", None, index[id(cls.weight)]))
        if cls.bias is not None:
            self.grad_fields.append((M, "g_head_b", None, index[id(cls.bias)]))
        nb = 0
        self.block_modules = []
        for q, seq in enumerate(self.sequences):
            sq = M.seq[q]
            sq.level = 0 if q == 0 else (q if q <= S else q - S - 1)
            sq.depth, sq.first_block, sq.k = len(seq.blocks), nb, seq.neighbours
            for blk in seq.blocks:
                a = blk.attn
                sq.c, sq.g = a.embed_channels, a.groups
                params, blk_bns = _block.block_params(blk)
                mb = M.block[nb]
                for i, p in enumerate(params):
                    mb.param[i] = ptr(p)
                    if p is not None:
                        self.grad_fields.append((mb.gparam, None, i, index[id(p)]))
                for i, bn in enumerate(blk_bns):
                    mb.run_mean[i], mb.run_var[i] = buf(bn, "running_mean"), buf(bn, "running_var")
                    mb.batches[i] = buf(bn, "num_batches_tracked") if bn.track_running_stats else None
                bns.extend(blk_bns)
                self.block_modules.append(blk)
                nb += 1
        M.num_blocks = nb
        b0 = bns[0]
        M.eps, M.momentum = float(b0.eps), float(b0.momentum if b0.momentum is not None else 0.1)
        self.uniform_bn = all(bn.affine and bn.momentum is not None and bn.eps == b0.eps and bn.momentum == b0.momentum
                              and (bn.running_mean is None) == (b0.running_mean is None) for bn in bns)
        self.has_running = b0.running_mean is not None
        self.static_ok = (self.uniform_bn and nb <= MAX_BLOCKS and S <= MAX_STAGES and len(self.grad_fields) == len(self.params)
                          and all(p.dtype == torch.float32 and p.is_cuda and p.is_contiguous() for p in self.params)
                          and all(_block.plan(blk).static_core for blk in self.block_modules)
                          and all(blk.attn.attn_drop_rate == 0.0
                                  or _gva.dropout_supported(blk.attn.embed_channels, blk.attn.groups, seq.neighbours)
                                  for seq in self.sequences for blk in seq.blocks)
                          and all(_gva.supported(blk.attn.embed_channels, blk.attn.groups, seq.neighbours)
                                  for seq in self.sequences for blk in seq.blocks)
                          and isinstance(model.seg_head, torch.nn.Sequential) and M.embed.cout % 4 == 0)
        self.M = M
        self.key = _block._pointer_key(self.tensor_slots)
        self.droppath = [[b.drop_path.drop_prob if hasattr(b.drop_path, "drop_prob") else 0.0 for b in seq.blocks]
                         for seq in self.sequences]
        self._grad_buf = None   # persistent flat gradient buffer + its per-parameter views (direct mode / cached views)
        self._grad_views = None
        self._keep_cache = {}

    # -- per step ------------------------------------------------------------------------------------------------
    def fill_prefix(self, lv):
        """Level 0 and seq 0 alone: what ptv2_model_forward_prefix_hip_launcher reads (the other levels are not known yet)."""
        M = self.M
        L = M.level[0]
        L.n, L.b = lv.coord.shape[0], lv.offset.numel()
        L.coord, L.offset = lv.coord.data_ptr(), lv.offset.data_ptr()
        L.order = L.idx_ptr = L.cluster = L.up_idx = L.up_w = L.up_inv_ptr = L.up_inv_rows = None
        sq = M.seq[0]
        idx = lv.neighbours(sq.k)
        mu, cov = _gva._pos_moments(_gva._HipImpl, lv.coord, idx)
        sq.idx, sq.mu, sq.cov, sq.inv_ptr, sq.inv_rows = idx.data_ptr(), mu.data_ptr(), cov.data_ptr(), None, None
        return [idx, mu, cov]

    def geometry_worker(self, device):
        """The helper thread the pipelined forward runs its native geometry call on (one per runtime; the call releases the
        interpreter lock and spends most of its time blocked in the poolings' read-backs)."""
        ex = self.__dict__.get("_geo_worker")
        if ex is None or self.__dict__.get("_geo_worker_device") != device:
            from concurrent.futures import ThreadPoolExecutor

            ex = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ao_amd-geometry",
                                    initializer=torch.cuda.set_device, initargs=(device,))
            self.__dict__["_geo_worker"], self.__dict__["_geo_worker_device"] = ex, device
        return ex

    def side_stream(self, device):
        """The stream the pipelined forward builds the deeper levels' geometry on (one per runtime and device)."""
        st = self.__dict__.get("_side")
        if st is None or st.device != device:
            st = torch.cuda.Stream(device, priority=-1 if os.environ.get("AO_AMD_SIDE_PRIORITY") == "high" else 0)
            self.__dict__["_side"] = st
        return st

    def fill_geometry_raw(self, geo):
        """fill_geometry from a geometry.RawSceneGeometry: sizes and addresses straight from the native call's struct."""
        M, S, G = self.M, self.S, geo.G
        lv0 = geo.lv0
        for i in range(S + 1):
            L, Gi = M.level[i], G.level[i]
            L.n, L.b = Gi.n, G.b
            if i == 0:
                L.coord, L.offset = lv0.coord.data_ptr(), lv0.offset.data_ptr()
            else:
                L.coord, L.offset = geo.addr(Gi.coord), geo.addr(Gi.offset)
            L.order, L.idx_ptr, L.cluster = geo.addr(Gi.order), geo.addr(Gi.idx_ptr), geo.addr(Gi.cluster)
            L.up_idx, L.up_w = geo.addr(Gi.up_idx), geo.addr(Gi.up_w)
            L.up_inv_ptr, L.up_inv_rows = geo.addr(Gi.up_inv_ptr), geo.addr(Gi.up_inv_rows)
        for q in range(len(self.sequences)):
            sq = M.seq[q]
            sq.idx, sq.mu, sq.cov, sq.inv_ptr, sq.inv_rows = geo.table(sq.level, sq.k)
        return geo.tensors()

    def fill_geometry(self, geo):
        if hasattr(geo, "G"):
            return self.fill_geometry_raw(geo)
        M, S = self.M, self.S
        for i, lv in enumerate(geo.levels):
            L = M.level[i]
            L.n, L.b = lv.coord.shape[0], lv.offset.numel()
            L.coord, L.offset = lv.coord.data_ptr(), lv.offset.data_ptr()
            L.order = _lib.ptr(lv.order32) or None
            L.idx_ptr = _lib.ptr(lv.idx_ptr32) or None
            L.cluster = _lib.ptr(lv.cluster) or None
            L.up_idx, L.up_w = _lib.ptr(lv.up_idx) or None, _lib.ptr(lv.up_weight) or None
            if lv.up_idx is not None:
                inv_ptr, inv_rows = _gva.inverse_table(lv.up_idx)
                L.up_inv_ptr, L.up_inv_rows = inv_ptr.data_ptr(), inv_rows.data_ptr()
            else:
                L.up_inv_ptr = L.up_inv_rows = None
        keep = []
        for q, seq in enumerate(self.sequences):
            sq = M.seq[q]
            lv = geo.levels[sq.level]
            idx = lv.neighbours(sq.k)
            sq.idx = idx.data_ptr()
            mu, cov = _gva._pos_moments(_gva._HipImpl, lv.coord, idx)
            inv_ptr, inv_rows = _gva.inverse_table(idx)
            sq.mu, sq.cov, sq.inv_ptr, sq.inv_rows = mu.data_ptr(), cov.data_ptr(), inv_ptr.data_ptr(), inv_rows.data_ptr()
            keep += [idx, mu, cov, inv_ptr, inv_rows]
        return keep

    def draw_droppath(self, geo, device, q_from=0, q_to=None):
        """Per-point DropPath factors (timm DropPath on an (N,C) tensor, point_transformer_v2m2_base.py:160-162,175) of
        every block of the sequences q_from .. q_to in ONE draw: Bernoulli(keep_b) / keep_b for the rows of block b, 0-rate
        blocks skipped.  (The pipelined forward draws for seq 0 first and for the others once their levels' sizes are known.)"""
        if q_to is None:
            q_to = len(self.droppath) - 1
        sizes = tuple(geo.sizes) if hasattr(geo, "sizes") else tuple(lv.coord.shape[0] for lv in geo.levels)
        key = (sizes[: 1 + max(self.M.seq[q].level for q in range(q_from, q_to + 1))], q_from, q_to)
        entry = self._keep_cache.get(key)
        if entry is None:
            spans, probs, off = [], [], 0
            for q in range(q_from, q_to + 1):
                n = sizes[self.M.seq[q].level]
                nb = self.M.seq[q].first_block
                for r in self.droppath[q]:
                    if r > 0.0:
                        spans.append((nb, off, n))
                        probs.append(torch.full((n,), 1.0 - r, dtype=torch.float32))
                        off += n
                    nb += 1
            keep = torch.cat(probs).to(device) if probs else None
            entry = (spans, keep)
            # scenes change size every batch: keep the latest layout of each range only
            self._keep_cache = {k: v for k, v in self._keep_cache.items() if k[1:] != key[1:]}
            self._keep_cache[key] = entry
        spans, keep = entry
        for q in range(q_from, q_to + 1):
            sq = self.M.seq[q]
            for mb in self.M.block[sq.first_block: sq.first_block + sq.depth]:
                mb.rowscale = None
        if keep is None:
            return None
        scales = torch.bernoulli(keep).div_(keep)
        base = scales.data_ptr()
        for nb, off, n in spans:
            self.M.block[nb].rowscale = base + 4 * off
        return scales

    def draw_droppath_bound(self, n0, device):
        """draw_droppath for all blocks before the deeper levels' sizes are known: every dropping block gets n0 factors (its
        level has at most n0 rows and reads the first n of them)."""
        key = ("bound", n0)
        entry = self._keep_cache.get(key)
        if entry is None:
            slots, probs = [], []
            nb = 0
            for rates in self.droppath:
                for r in rates:
                    if r > 0.0:
                        slots.append(nb)
                        probs.append(1.0 - r)
                    nb += 1
            keep = (torch.tensor(probs, dtype=torch.float32, device=device).repeat_interleave(n0) if probs else None)
            entry = (slots, keep)
            self._keep_cache = {k: v for k, v in self._keep_cache.items() if k[0] != "bound"}
            self._keep_cache[key] = entry
        slots, keep = entry
        for mb in self.M.block[: self.M.num_blocks]:
            mb.rowscale = None
        if keep is None:
            return None
        scales = torch.bernoulli(keep).div_(keep)
        base = scales.data_ptr()
        for slot, nb in enumerate(slots):
            self.M.block[nb].rowscale = base + 4 * slot * n0
        return scales

    def grad_buffer(self, device, fresh):
        """(flat buffer, per-parameter views).  "direct" mode reuses one persistent pair step after step (`.grad` then
        aliases a buffer that the next backward overwrites -- the mode's documented contract); `fresh` allocates a
        private pair: always in "autograd" mode (callers of torch.autograd.grad may hold the returned tensors), and
        for gradient accumulation over several backwards."""
        if fresh or self._grad_buf is None or self._grad_buf.device != device:
            flat = torch.zeros(self.total, dtype=torch.float32, device=device)
            views = self._views_of(flat)
            if fresh:
                return flat, views
            self._grad_buf, self._grad_views = flat, views
        return self._grad_buf, self._grad_views

    def _views_of(self, flat):
        """One view of `flat` per parameter.  "autograd" mode needs NEW tensor objects every backward (AccumulateGrad keeps a
        gradient it alone references and clones one that is referenced elsewhere): one split call produces the slots, only the
        parameters with more than one dimension (or a padded slot) take a python-level view on top -- 0.8 ms against 2.9 ms for
        a slice + view per parameter (492 parameters, measured on the build host)."""
        plan = self.__dict__.get("_view_plan")
        if plan is None:
            sizes = [b - a for a, b in zip(self.offsets, list(self.offsets[1:]) + [self.total])]
            if self.offsets and self.offsets[0] == 0 and all(sz >= p.numel() for sz, p in zip(sizes, self.params)):
                fix = [(i, p.numel() if p.numel() != sz else None, tuple(p.shape) if p.dim() != 1 else None)
                       for i, (p, sz) in enumerate(zip(self.params, sizes)) if p.dim() != 1 or p.numel() != sz]
                plan = (sizes, fix)
            else:
                plan = False
            self.__dict__["_view_plan"] = plan
        if plan is False:
            return [flat[off:off + p.numel()].view(p.shape) for p, off in zip(self.params, self.offsets)]
        sizes, fix = plan
        views = list(torch.split_with_sizes(flat, sizes))
        for i, numel, shape in fix:
            t = views[i]
            if numel is not None:
                t = t.narrow(0, 0, numel)
            if shape is not None:
                t = t.view(shape)
            views[i] = t
        return views

    def point_grads(self, flat):
        base = flat.data_ptr()
        offs = self.offsets
        for obj, field, i, pi in self.grad_fields:
            if field is None:
                obj[i] = base + 4 * offs[pi]
            else:
                setattr(obj, field, base + 4 * offs[pi])


def runtime(model, check=True):
    """The model's _Runtime, rebuilt when a parameter or buffer has moved.  check=False: the caller runs right behind a
    `supported(model, feat)` of the same forward, which has just compared all 841 tensor addresses (0.2 ms of the launching
    thread per comparison: three per forward before)."""
    rt = model.__dict__.get("_ao_runtime")
    if rt is not None and not check:
        return rt
    if rt is not None:
        try:
            fresh = rt.key == _block._pointer_key(rt.tensor_slots)
        except (KeyError, AttributeError):
            fresh = False
        if fresh:
            return rt
    rt = _Runtime(model)
    model.__dict__["_ao_runtime"] = rt
    return rt


def supported(model, feat):
    if os.environ.get("AO_AMD_MODEL", "native") != "native" or os.environ.get("AO_AMD_BLOCK", "native") != "native":
        return False
    if os.environ.get("AO_AMD_GVA", "fused") != "fused":
        return False
    if not (feat.is_cuda and feat.dim() == 2 and feat.dtype in (torch.float32, torch.bfloat16, torch.float16) and feat.shape[0] >= 2):
        return False
    # the native backward forms no gradient for `feat` (model.hip stops at the patch embedding's weights): a caller whose
    # features come out of a trainable upstream module takes the stage-wise python path, which does
    if feat.requires_grad and torch.is_grad_enabled():
        return False
    if feat.shape[1] != model.in_channels:
        raise ValueError("ao_amd: feat has %d channels, the model was built with in_channels=%d" % (feat.shape[1], model.in_channels))
    rt = runtime(model)
    if not rt.static_ok or (not model.training and not rt.has_running):
        return False
    return True  # (enable_checkpoint is honoured by the native runtime: ptv2_model.checkpoint)


def geometry_supported(geo):
    """every level needs >= 2 rows (training-mode BatchNorm, ptv2_block args_ok): tiny clouds take the python path"""
    if hasattr(geo, "sizes"):
        return all(n >= 2 for n in geo.sizes)
    return all(int(lv.coord.shape[0]) >= 2 for lv in geo.levels)


class _Pipelined:
    """Stands in for the SceneGeometry of a forward that builds it itself, pipelined with the level-0 prefix."""

    def __init__(self, model, coord, offset):
        self.model, self.coord, self.offset = model, coord, offset


def _side_tensors(geo):
    from .parallel import _geometry_tensors
    return _geometry_tensors(geo)


class _NativeModel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *args):
        try:
            return _NativeModel._forward(ctx, *args)
        except BaseException:
            # the helper thread may still be inside the native geometry call, enqueuing on the side stream: let it finish (its
            # error, if any, is the consequence, not the cause) before the exception unwinds -- the next forward would queue
            # behind it on the single-worker executor and find the side stream in an unknown state
            fut = ctx.__dict__.pop("_geo_fut", None) if hasattr(ctx, "__dict__") else None
            if fut is not None:
                try:
                    fut.result()
                except Exception:
                    pass
            raise

    @staticmethod
    def _forward(ctx, feat, anchor, rt, geo, training, mode, bf16, *params):
        feat = feat.contiguous()
        dev = feat.device
        L = _lib.lib()
        M = rt.M
        M.training, M.matmul_bf16 = int(training), int(bf16)
        # activation checkpointing (reference :169-171: only while training with gradients enabled)
        M.checkpoint = int(bool(rt.want_checkpoint))  # decided by forward() below (grad mode is off inside this function)
        ctx.checkpoint = M.checkpoint
        # attention dropout: one fresh mask seed per Block and step (the backward re-evaluates the mask from it)
        for mb, blk in zip(M.block[: M.num_blocks], rt.block_modules):
            rate = blk.attn.attn_drop_rate if training else 0.0
            mb.attn_drop_p, mb.attn_drop_seed = (float(rate), _gva.next_drop_seed()) if rate > 0.0 else (0.0, 0)
        ctx.attn_drop = [(mb.attn_drop_p, mb.attn_drop_seed) for mb in M.block[: M.num_blocks]]
        n0 = feat.shape[0]
        M.feat, M.logits = feat.data_ptr(), None
        M.saved, M.saved_bytes, M.saved0, M.saved0_bytes = None, 0, None, 0
        saved0, scales0, ev_all, fut, job = None, None, None, None, None
        if isinstance(geo, _Pipelined):
            # The geometry is built HERE, pipelined with the network (no prefetcher thread, no `geometry=` batch key: the
            # reference trainer's `model(input_dict)`, pointcept/engines/train_sam_pp2s.py:181).  Only the grid poolings have
            # data-dependent sizes (one 4-byte read-back each); nothing at level 0 depends on them.  So: the level-0 tables and
            # the level-0 PREFIX of the network (patch embedding: ~1.5 ms of GPU work at 120 k points) are enqueued first, on the
            # caller's stream; the poolings, the deeper levels' tables and the inverse tables follow on a side stream, where
            # their read-backs wait for the geometry kernels only -- the host blocks there while the GPU is busy with the
            # prefix; then the rest of the network is enqueued behind an event.
            from .geometry import NativeGeometryJob, begin_geometry, finish_geometry, native_finish_supported
            req, model = geo, geo.model
            main = torch.cuda.current_stream(dev)
            side = rt.side_stream(dev)
            # (materialised BEFORE the event: a non-contiguous coord view -- feat[:, :3] -- or an int64 offset becomes a copy kernel
            # on this stream, which the side stream's first pooling must not overtake)
            coord0, offset0 = req.coord.contiguous(), req.offset.int().contiguous()
            ev_in = torch.cuda.Event()
            ev_in.record(main)  # coord / offset may have been produced on this stream (the trainer's H2D copies)
            st = begin_geometry(coord0, offset0, model.grid_sizes, model.geometry_neighbours(),
                                interp=model.unpool_backend == "interp")
            ev_l0 = torch.cuda.Event()
            ev_l0.record(main)  # the level-0 tables exist (the inverse tables on the side stream read them)
            lv0 = st.geo.levels[0]
            ev_fwd, ev_all = torch.cuda.Event(), torch.cuda.Event()
            native = native_finish_supported(st)
            for t in (lv0.coord, lv0.offset, *lv0.knn.values()):  # made on the caller's stream, read on the side stream
                t.record_stream(side)
            if native:
                # one native call for everything behind the first pooling (csrc/scene.hip), on a helper thread of the runtime: it
                # blocks in the poolings' read-backs while THIS thread captures and launches the prefix; the struct's progress
                # flags tell this thread when the sizes are in
                with torch.cuda.stream(side):
                    side.wait_event(ev_in)
                    ev_fwd.record(side)  # (creates the handle the launcher re-records)
                    job = NativeGeometryJob(st, fwd_ready_event=ev_fwd, knn0_event=ev_l0)
                fut = rt.geometry_worker(dev).submit(job.run)
                ctx._geo_fut = fut
            keep0 = rt.fill_prefix(lv0)
            for t in keep0:
                t.record_stream(side)
            # DropPath factors of EVERY block in one draw, now: rows sized by level 0 (no level has more), so that no random-number
            # launch sits between the geometry's last read-back and the rest of the network
            scales0 = rt.draw_droppath_bound(lv0.coord.shape[0], dev) if training else None
            need0 = L.ptv2_model_prefix_saved_bytes(ctypes.addressof(M))
            if need0 == 0:
                raise RuntimeError("ao_amd: ptv2_model prefix rejected by the native runtime")
            saved0 = torch.empty(need0, dtype=torch.uint8, device=dev)
            M.saved0, M.saved0_bytes = saved0.data_ptr(), saved0.numel()
            ws = _lib.workspace(L.ptv2_model_prefix_workspace_bytes(ctypes.addressof(M)), dev)
            rc = L.ptv2_model_forward_prefix_hip_launcher(ctypes.addressof(M), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, "ptv2_model_forward_prefix_hip_launcher")
            if native:
                job.wait_sizes(fut)          # the last read-back is in: every size and address is final
                geo = job.geometry()
            else:
                def before_inverse():
                    ev_fwd.record(side)      # everything the forward's rest needs is enqueued
                    side.wait_event(ev_l0)

                with torch.cuda.stream(side):
                    side.wait_event(ev_in)
                    geo = finish_geometry(st, before_inverse=before_inverse)
                    ev_all.record(side)
            if not geometry_supported(geo):
                raise RuntimeError("ao_amd: a level of this batch has fewer than 2 points; training-mode BatchNorm needs 2 "
                                   "(the reference's nn.BatchNorm1d raises here as well)")
            keep = rt.fill_geometry(geo)
            scales = None
        else:
            keep = rt.fill_geometry(geo)
            scales = rt.draw_droppath(geo, dev) if training else None
        logits = torch.empty((n0, M.num_classes), dtype=torch.float32, device=dev)
        M.logits = logits.data_ptr()
        need = L.ptv2_model_saved_bytes(ctypes.addressof(M))
        if need == 0:
            raise RuntimeError("ao_amd: ptv2_model rejected by the native runtime (ptv2_model_saved_bytes == 0)")
        saved = torch.empty(need, dtype=torch.uint8, device=dev)
        M.saved, M.saved_bytes = saved.data_ptr(), saved.numel()
        ws = _lib.workspace(L.ptv2_model_workspace_bytes(ctypes.addressof(M)), dev)
        if saved0 is not None:
            if fut is not None:
                job.wait_fwd_recorded(fut)   # (the deeper levels' tables are enqueued, the event re-recorded: usually long since)
            main.wait_event(ev_fwd)
            for t in _side_tensors(geo):  # allocated on the side stream, consumed on the caller's
                t.record_stream(main)
            rc = L.ptv2_model_forward_rest_hip_launcher(ctypes.addressof(M), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, "ptv2_model_forward_rest_hip_launcher")
            if fut is not None:  # the helper thread's call has enqueued the inverse tables by now (the backward waits for them)
                fut.result()
                job.check()
                ev_all.record(side)
        else:
            rc = L.ptv2_model_forward_hip_launcher(ctypes.addressof(M), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, "ptv2_model_forward_hip_launcher")
        ctx.rt, ctx.geo, ctx.keep, ctx.mode, ctx.training, ctx.bf16 = rt, geo, (keep, (scales, scales0), feat, saved, saved0), mode, training, bf16
        ctx.ev_all = ev_all  # (the backward waits for the inverse tables; without one, their inputs carry record_stream marks)
        ctx.rowscale_ptrs = [mb.rowscale for mb in M.block[: M.num_blocks]]
        return logits

    @staticmethod
    def backward(ctx, g_logits):
        rt = ctx.rt
        if ctx.keep is None:
            raise RuntimeError("ao_amd: the native PT-v2m2 runtime releases its saved activations at the end of the backward "
                               "(retain_graph / a second backward through the same forward is not supported)")
        keep, scales, feat, saved, saved0 = ctx.keep
        dev = feat.device
        if ctx.ev_all is not None:  # pipelined forward: the inverse tables were built on the side stream
            torch.cuda.current_stream(dev).wait_event(ctx.ev_all)
        L = _lib.lib()
        M = rt.M
        rt.fill_geometry(ctx.geo)  # the struct is shared between calls: restore this call's tables
        M.training, M.matmul_bf16, M.checkpoint = int(ctx.training), int(ctx.bf16), int(ctx.checkpoint)
        for mb, rs, (dp, ds) in zip(M.block[: M.num_blocks], ctx.rowscale_ptrs, ctx.attn_drop):
            mb.rowscale = rs
            mb.attn_drop_p, mb.attn_drop_seed = dp, ds
        M.feat, M.logits = feat.data_ptr(), None
        M.logits = g_logits.data_ptr()  # unused by the backward; keeps the struct valid
        M.saved, M.saved_bytes = saved.data_ptr(), saved.numel()
        M.saved0, M.saved0_bytes = (saved0.data_ptr(), saved0.numel()) if saved0 is not None else (None, 0)
        # parallel.FlatGradSync(mode="flat2"): an event the launcher records once the head + decoder gradients are final
        owner = rt.model_ref()
        ev = owner.__dict__.get("native_decoder_done_event") if owner is not None else None
        M.decoder_done_event = ev.cuda_event if ev is not None else None
        if ev is not None:
            owner.__dict__["native_decoder_done_count"] = owner.__dict__.get("native_decoder_done_count", 0) + 1
        direct = ctx.mode == "direct"
        accumulate = direct and rt.params[0].grad is not None
        if owner is not None:  # parallel.FlatGradSync: the early chunk may only leave from the launcher's own, unaccumulated buffer
            owner.__dict__["native_last_backward_accumulated"] = bool(accumulate or not direct)
        flat, views = rt.grad_buffer(dev, fresh=accumulate or not direct)
        if flat is rt._grad_buf:
            flat.zero_()
        rt.point_grads(flat)
        g_logits = g_logits.contiguous().float()
        ws = _lib.workspace(L.ptv2_model_workspace_bytes(ctypes.addressof(M)), dev)
        rc = L.ptv2_model_backward_hip_launcher(ctypes.addressof(M), g_logits.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "ptv2_model_backward_hip_launcher")
        # the activation arena (≈3 GB at 120 k points) goes back to the allocator now, not when the loss tensor that holds this
        # graph is dropped -- a training loop keeps the previous step's loss / output dict alive into the next forward, which
        # then needed a SECOND arena (and the allocator a multi-gigabyte hipMalloc in the middle of the second step)
        ctx.keep = ctx.geo = None
        if owner is not None and owner.__dict__.get("_ao_ddp_native_sync"):
            # under a DistributedDataParallel wrapper that leaves the parameters to us (model.parallel_ddp_ignore): the average
            # over the ranks as ONE all-reduce of the flat buffer, enqueued behind the backward; the caller's stream waits for it
            import torch.distributed as dist

            if dist.is_initialized():
                if dist.get_backend() == "nccl":  # RCCL averages in the collective
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True).wait()
                else:  # (gloo, the two-ranks-on-one-device tests: no AVG)
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                    flat.mul_(1.0 / dist.get_world_size())
        if direct:
            if accumulate:
                torch._foreach_add_([p.grad for p in rt.params], views)
            else:
                for p, v in zip(rt.params, views):
                    p.grad = v
            return (None,) * (7 + len(rt.params))
        return (None,) * 7 + tuple(views)


def pipelined_ok(model):
    """Whether forward(model, data, None) may build the geometry itself, pipelined with the level-0 prefix: training mode
    (a level of < 2 points is an error there, as in the reference; eval mode takes such batches through the python path,
    which has to see the sizes first), no activation checkpointing (one shared saved region), AO_AMD_PIPELINE != 0."""
    if os.environ.get("AO_AMD_PIPELINE", "1") == "0" or not model.training:
        return False
    rt = runtime(model, check=False)
    return not (torch.is_grad_enabled() and any(b.enable_checkpoint for b in rt.block_modules)) and rt.M.seq[0].depth > 0


def forward(model, data_dict, geo):
    """PointTransformerV2.forward on the native runtime (call `supported` first).  geo None: the geometry is built inside,
    pipelined with the network's level-0 prefix (call `pipelined_ok` first)."""
    rt = runtime(model, check=False)
    feat = data_dict["feat"]
    if geo is None:
        geo = _Pipelined(model, data_dict["coord"], data_dict["offset"].int())
    training = model.training or not rt.has_running
    mode = getattr(model, "native_param_grads", "autograd")
    bf16 = matmul_bf16()
    # activation checkpointing as the reference applies it (:169-171): only while training with gradients enabled
    rt.want_checkpoint = bool(training and torch.is_grad_enabled() and any(b.enable_checkpoint for b in rt.block_modules))
    with torch.autocast("cuda", enabled=False):  # activations, statistics, softmax and accumulation stay fp32
        if not torch.is_grad_enabled():
            return _NativeModel.apply(feat.float(), None, rt, geo, training, mode, bf16)
        if mode == "direct":
            anchor = model.__dict__.get("_ao_anchor")
            if anchor is None or anchor.device != feat.device:
                anchor = torch.zeros((), device=feat.device, requires_grad=True)
                model.__dict__["_ao_anchor"] = anchor
            return _NativeModel.apply(feat.float(), anchor, rt, geo, training, mode, bf16)
        return _NativeModel.apply(feat.float(), None, rt, geo, training, mode, bf16, *rt.params)


def matmul_bf16():
    """Under torch.autocast (the reference trainer's `enable_amp`, pointcept/engines/train_sam_pp2s.py:178-180) the
    nn.Linear products run on the bf16 matrix cores: operands rounded to bf16, fp32 accumulation -- autocast's own
    arithmetic for Linear.  Everything else (BatchNorm statistics, softmax, coordinates, activations in memory) stays
    fp32, as autocast keeps it.  A float16 autocast (the reference's default on CUDA) maps to the same bf16 operands:
    gfx950's fp16 and bf16 MFMA run at one rate and bf16 needs no loss scaling."""
    return bool(torch.is_autocast_enabled("cuda")) and os.environ.get("AO_AMD_AUTOCAST_MATMUL", "bf16") == "bf16"
