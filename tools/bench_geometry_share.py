#!/usr/bin/env python
"""How much of the step is the scene geometry?  The bench step with (a) the geometry prefetched on the side stream
(what bench.py times; `prefetch_early`: enqueued behind the forward instead of behind the backward), (b) the geometry built in line on the compute stream, (c) NO geometry work at all (one prebuilt
geometry reused: not a valid step, a lower bound for everything else), (d) the geometry alone.
usage: tools/bench_geometry_share.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ao_amd.ptv2 as ptv2
from ao_amd import synth
from ao_amd.ptv2 import parallel
from ao_amd.ptv2.optim import FlatAdamW

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
if os.environ.get("AO_AMD_MAIN_STREAM"):  # compute on a non-blocking stream instead of the null stream
    torch.cuda.set_stream(torch.cuda.Stream())
b = synth.scene_batch([0], point_max=120000, room=1)
data = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
torch.manual_seed(0)
seg = ptv2.DefaultSegmentor(ptv2.S3DIS_BACKBONE).to(dev).train()
seg.backbone.native_param_grads = "direct"
opt = FlatAdamW(seg.parameters(), lr=0.006, weight_decay=0.05)
pre = parallel.GeometryPrefetcher(seg.backbone, dev)
with torch.no_grad():
    fixed = seg.backbone.geometry(data["coord"], data["offset"])


def run(mode):
    if mode.startswith("prefetch"):
        pre.start(data["coord"], data["offset"])

    def step():
        if mode.startswith("prefetch"):
            geo = pre.take()
        elif mode == "inline":
            with torch.no_grad():
                geo = seg.backbone.geometry(data["coord"], data["offset"])
        else:
            geo = fixed
        if mode != "geometry_only":
            loss = seg(dict(data, geometry=geo))["loss"]
            if mode == "prefetch_early":  # enqueued right behind the forward instead of behind the backward
                pre.start(data["coord"], data["offset"])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            flat = opt.flatten_grads()
        if mode == "prefetch":
            pre.start(data["coord"], data["offset"])
        if mode == "geometry_only":
            with torch.no_grad():
                seg.backbone.geometry(data["coord"], data["offset"])
        else:
            opt.step(flat_grad=flat)

    if mode.startswith("eager"):
        # the model issued eagerly with the backward recording an event once the decoder's gradients are done (= in front of
        # the encoder's deep levels); a side load either at the start of the step or behind that event
        ev = torch.cuda.Event()
        seg.backbone.__dict__["native_decoder_done_event"] = ev
        where, _, load = mode[6:].partition(":")
        mode = "none"

        def side():
            with torch.no_grad(), torch.cuda.stream(pre.stream):
                if where == "gate":
                    pre.stream.wait_event(ev)
                if load == "inverse":
                    for t in tables:
                        t._ao_inverse = None
                    gva.inverse_tables(tables)
                elif load == "knn":
                    knn_query_dist2(16, data["coord"], data["offset"].int())
                    knn_query_dist2(8, data["coord"], data["offset"].int())
                elif load == "wide":
                    big.add_(1.0)

        def step():  # noqa: F811
            if where == "start":
                side()
            loss = seg(dict(data, geometry=fixed))["loss"]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if where == "gate":
                side()
            opt.step(flat_grad=opt.flatten_grads())

    side_mode = None
    if mode.startswith("none+"):  # the fixed geometry + a synthetic load on the side stream (what part of the geometry costs?)
        side_mode, mode = mode[5:], "none"
        inner = step

        def step():  # noqa: F811
            with torch.no_grad(), torch.cuda.stream(pre.stream):
                if side_mode.startswith("tiny"):     # N launches of nothing
                    for _ in range(int(side_mode[4:])):
                        tiny.add_(1.0)
                elif side_mode == "inverse":         # the five launches of the inverse tables (atomics)
                    for t in tables:
                        t._ao_inverse = None
                    gva.inverse_tables(tables)
                elif side_mode == "knn":             # the self tables of level 0 (grid build + query)
                    knn_query_dist2(16, data["coord"], data["offset"].int())
                    knn_query_dist2(8, data["coord"], data["offset"].int())
                elif side_mode == "wide":            # one wide bandwidth-bound launch of ~0.3 ms (1 GB read + write)
                    big.add_(1.0)
            inner()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if mode.startswith("prefetch"):
        pre.take()
    seg.backbone.__dict__.pop("native_decoder_done_event", None)
    return 1e3 * (time.perf_counter() - t0) / steps


from ao_amd.ptv2 import gva  # noqa: E402
from ao_amd.ptv2.geometry import knn_query_dist2  # noqa: E402

tiny = torch.zeros(64, device=dev)
big = torch.zeros(128 << 20, device=dev)
tables = [lv.up_idx for lv in fixed.levels if lv.up_idx is not None] + [i for lv in fixed.levels for i in lv.knn.values()]
tables = [t.clone() for t in tables]  # (copies: the model keeps using the tables of `fixed` and their inverses)
modes = ("prefetch", "prefetch_early", "inline", "none", "geometry_only", "prefetch", "prefetch_early")
if os.environ.get("AO_AMD_SIDE_LOADS"):
    modes = ("none", "prefetch", "none+tiny50", "none+tiny100", "none+tiny200", "none+inverse", "none+knn", "none+wide", "none", "prefetch")
if os.environ.get("AO_AMD_SIDE_GATE"):
    modes = ("none", "eager", "eager+start:inverse", "eager+gate:inverse", "eager+start:knn", "eager+gate:knn", "eager+start:wide",
             "eager+gate:wide", "eager")
for mode in modes:
    print("%-14s %.3f ms/step" % (mode, run(mode)))
