"""Data-parallel step plumbing shared by bench.py and the CPU (gloo) tests.

The hot path shards by scene: one process per GPU, whole scenes per rank, no data-path
collective; the only exchange is the gradient all-reduce, which torch's DistributedDataParallel buckets and
overlaps with backward over RCCL (backend "nccl" on ROCm) -- the reference does exactly this
(pointcept/engines/defaults.py:22-43, train_sam_pp2s.py:209-213).
"""
import os
import time

import torch
import torch.distributed as dist


def rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def scene_seeds(rank, scenes_per_rank):
    """Scenes owned by `rank` (disjoint across ranks; weak scaling: fixed work per rank)."""
    return [rank * scenes_per_rank + i for i in range(scenes_per_rank)]


def wrap_ddp(module, device):
    """DDP with the reference's settings (broadcast_buffers=False: BatchNorm statistics stay per process)."""
    ids = [device.index] if device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(module, device_ids=ids, broadcast_buffers=False,
                                                     gradient_as_bucket_view=True)


class FlatGradSync:
    """Gradient averaging as ONE all-reduce of one flat buffer after backward (RCCL over xGMI on GPUs).

    DistributedDataParallel copies every parameter gradient into its buckets with one small kernel each; with the
    840 parameters of PT-v2m2 that is ~2.4 ms of a 17 ms step on one MI355X (measured with a 1-rank group), more
    than the 15.6 MB exchange itself costs on xGMI.  Here the gradients are flattened by one multi-tensor copy,
    reduced once, and written back by one multi-tensor copy.  Parameters are broadcast from rank 0 once, BatchNorm
    buffers stay per process (the reference runs DDP with broadcast_buffers=False and sync_bn=False)."""

    def __init__(self, module, force=False, mode="flat"):
        self.force = force  # run the flatten / all-reduce / write-back even in a 1-rank group (overhead measurement)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # mode "flat2": the flat buffer goes out in two chunks -- [head + decoder stages] as soon as the native backward
        # has finished them (an event the launcher records, ptv2_model.decoder_done_event), on a communication stream,
        # while the encoder's backward still runs; [patch embedding + encoder stages] after the backward.  For the
        # 45 MB ScanNet-cfg model the first chunk's ~0.5 ms of ring time disappears behind compute; at 15.6 MB (S3DIS
        # cfg) one collective is as good.  `module` must expose `.backbone` (a PointTransformerV2) or be one.
        self.split, self.event, self.comm = None, None, None
        if mode == "flat2":
            backbone = getattr(module, "backbone", module)
            names = [n for n, p in module.named_parameters() if p.requires_grad]
            first = next((i for i, n in enumerate(names) if "dec_stages." in n), None)
            if first is not None and first > 0 and torch.cuda.is_available() and self.params[0].is_cuda:
                off = 0
                for p in self.params[:first]:
                    off += (p.numel() + 3) // 4 * 4  # optim.FlatAdamW / native_model.grad_layout slots
                self.split = off
                self.event = torch.cuda.Event()
                self.event.record()  # creates the handle the launcher re-records
                self.comm = torch.cuda.Stream(self.params[0].device)
                backbone.__dict__["native_decoder_done_event"] = self.event
                self.backbone, self.seen = backbone, backbone.__dict__.get("native_decoder_done_count", 0)
        if self.world > 1:
            flat = torch._utils._flatten_dense_tensors([p.data for p in self.params])
            dist.broadcast(flat, 0)
            for p, f in zip(self.params, torch._utils._unflatten_dense_tensors(flat, [p.data for p in self.params])):
                p.data.copy_(f)

    def sync(self):
        """Average `.grad` over the ranks (call between backward and the optimizer step)."""
        if self.world == 1 and not self.force:
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        flat = torch._utils._flatten_dense_tensors(grads)
        dist.all_reduce(flat)
        flat.div_(self.world)
        torch._foreach_copy_(grads, list(torch._utils._unflatten_dense_tensors(flat, grads)))

    def reduce_flat(self, flat):
        """All-reduce (sum) an already flat gradient buffer in place (optim.FlatAdamW.flatten_grads()); the caller
        folds the 1 / world into its update (FlatAdamW.step(flat_grad=..., grad_scale=1 / world))."""
        if self.world > 1 or self.force:
            # two chunks only when THIS backward went through the native launcher (which re-recorded the event)
            recorded = self.split is not None and self.backbone.__dict__.get("native_decoder_done_count", 0) != self.seen
            if recorded:
                self.seen = self.backbone.__dict__["native_decoder_done_count"]
                # ... and only when `flat` IS the buffer the launcher wrote, untouched since: with gradients delivered through
                # autograd, accumulated over several backwards, or a segmentor with parameters of its own, flatten_grads()
                # fills `flat` by a copy / add on the main stream AFTER the event, and the early chunk would go out stale
                rt = self.backbone.__dict__.get("_ao_runtime")
                direct = rt is not None and rt._grad_buf is not None and flat.data_ptr() == rt._grad_buf.data_ptr()
                recorded = direct and not self.backbone.__dict__.get("native_last_backward_accumulated", True)
            if recorded and 0 < self.split < flat.numel():
                main = torch.cuda.current_stream(flat.device)
                self.comm.wait_event(self.event)  # the most recent record: enqueued by the backward that just returned
                with torch.cuda.stream(self.comm):
                    dist.all_reduce(flat[self.split:])
                dist.all_reduce(flat[:self.split])
                main.wait_stream(self.comm)
            else:
                dist.all_reduce(flat)
        return 1.0 / self.world


def fence(device):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def timed_steps(step_fn, steps, device, points_this_rank, finish=None):
    """Run exactly `steps` steps bracketed by barrier+synchronize; returns (max elapsed over ranks,
    total points per step over all ranks, last loss).  `finish` (e.g. LogitBasket.flush) runs inside the timed
    region after the last step: work the steps left in flight counts."""
    fence(device)
    t0 = time.perf_counter()
    loss = None
    for _ in range(steps):
        loss = step_fn()
    if finish is not None:
        finish()
    fence(device)
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    pts = torch.tensor([float(points_this_rank)], device=device, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(pts, op=dist.ReduceOp.SUM)
    return float(t.item()), float(pts.item()), loss


def _geometry_tensors(geo):
    if hasattr(geo, "tensors"):  # geometry.RawSceneGeometry: one arena + level 0's tensors
        yield from geo.tensors()
        return
    for lv in geo.levels:
        for v in vars(lv).values():
            if torch.is_tensor(v):
                yield v
                for t in getattr(v, "_ao_inverse", ()) or ():
                    if torch.is_tensor(t):
                        yield t
        for idx in lv.knn.values():
            yield idx
            for attr in ("_ao_inverse", "_ao_pos_moments"):
                for t in getattr(idx, attr, ()) or ():
                    if torch.is_tensor(t):
                        yield t


class GeometryPrefetcher:
    """Builds the scene geometry of the NEXT batches (kNN tables, grid pooling, interpolation tables: functions of
    the coordinates only) on a side HIP stream while the current batch trains, the way a data loader
    overlaps host-side preparation.  The pooling has data-dependent output sizes, i.e. 4-byte read-backs; on the
    compute stream each of them drains the whole queue and leaves the GPU idle while the host catches up
    (profiles/r01: 3.7 ms of a 30 ms step).  On the side stream the read-backs wait for the geometry kernels only.

      pre.start(coord, offset)      enqueue the geometry of a coming batch (side stream); may be called several times ahead
      geo = pre.take()              the oldest one not yet taken: make the compute stream wait for it, hand it over
    """

    def __init__(self, backbone, device, threaded=False):
        self.backbone, self.device = backbone, device
        self.stream = torch.cuda.Stream(device)
        # threaded: the geometry is built by ONE host thread of its own, as a loader worker would, request after request (the
        # builds share the side stream and its scratch, so they must not interleave).  Building it is ~2 ms of host time per
        # 120 k-point scene, most of it blocked in the 4-byte read-backs; on the launching thread those come out of the time the
        # host is ahead of the GPU queue, which a busy host (other tenants on the box) eats first.  The launchers release the GIL
        # (ctypes), so the two threads overlap where it matters.  With two builds requested ahead (bench.py) the launching
        # thread finds the next geometry finished instead of waiting for the worker's read-backs.
        self.threaded, self.worker = threaded, None
        import collections
        self.done = collections.deque()  # (geo, event) or an exception, oldest first (not threaded)
        self.outstanding = 0
        if threaded:
            import queue
            self.requests, self.results = queue.Queue(), queue.Queue()

    def _build(self, coord, offset, ready):
        if ready is not None:  # event after which coord/offset are valid (e.g. the H2D copy of the loader)
            self.stream.wait_event(ready)
        with torch.no_grad(), torch.cuda.stream(self.stream):
            geo = self.backbone.geometry(coord, offset)
            done = torch.cuda.Event()
            done.record(self.stream)
        return geo, done

    def _loop(self):
        torch.cuda.set_device(self.device)
        while True:
            req = self.requests.get()
            if req is None:
                return
            try:
                self.results.put(self._build(*req))
            except BaseException as e:  # re-raised by take()
                self.results.put(e)

    def start(self, coord, offset, ready=None):
        self.outstanding += 1
        if not self.threaded:
            self.done.append(self._build(coord, offset, ready))
            return
        if self.worker is None:
            import threading
            self.worker = threading.Thread(target=self._loop, name="ao_amd-geometry", daemon=True)
            self.worker.start()
        self.requests.put((coord, offset, ready))

    @property
    def pending(self):
        return self.outstanding if self.outstanding else None

    def close(self):
        """Wait for the builds that are still running and stop the worker thread (call before the process winds down)."""
        if self.threaded and self.worker is not None:
            self.requests.put(None)
            self.worker.join()
            self.worker = None
            while not self.results.empty():
                self.results.get()
        self.done.clear()
        self.outstanding = 0

    def take(self):
        assert self.outstanding > 0, "GeometryPrefetcher.take() before start()"
        self.outstanding -= 1
        res = self.results.get() if self.threaded else self.done.popleft()
        if isinstance(res, BaseException):
            raise res
        geo, done = res
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        for t in _geometry_tensors(geo):  # allocated on the side stream, consumed on the compute stream
            t.record_stream(main)
        return geo
