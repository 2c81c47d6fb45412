#!/usr/bin/env python
"""What REAL's per-step logit basket costs on top of the training step (BASELINE.json configs[3]): the bench step with
(a) DefaultSegmentor, (b) DefaultSegmentorSAM_Image without a basket, (c) + LogitBasket.put with the scatter disabled,
(d) + the full basket, (e) the reference's blocking statement (two .cpu() copies per scene + numpy scatter).
usage: tools/bench_basket.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ao_amd.ptv2 as ptv2
from ao_amd import synth
from ao_amd.ptv2 import parallel
from ao_amd.ptv2.optim import FlatAdamW

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
b = synth.scene_batch([0], point_max=120000, room=1)
data = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
n = data["coord"].shape[0]
data["scene_id"] = ["Area_1/room_0.pth"]
data["instance"] = torch.randperm(2 * n, device=dev)[:n]
data["offset_host"] = [n]


def run(mode):
    torch.manual_seed(0)
    cls = ptv2.DefaultSegmentor if mode == "default" else ptv2.DefaultSegmentorSAM_Image
    seg = cls(ptv2.S3DIS_BACKBONE).to(dev).train()
    opt = FlatAdamW(seg.parameters(), lr=0.006, weight_decay=0.05)
    pre = parallel.GeometryPrefetcher(seg.backbone, dev)
    pre.start(data["coord"], data["offset"])
    basket = None
    if mode in ("copy_only", "full"):
        basket = ptv2.LogitBasket({"Area_1_room_0": 2 * n}, 13, device=dev, max_rows=n)
        if mode == "copy_only":
            basket._scatter = lambda *a: 0
    ref_basket = {"Area_1_room_0": np.full((2 * n, 13), -100.0, np.float32)}

    def step():
        out = seg(dict(data, geometry=pre.take()))
        if mode != "default":
            out, seg_dict = out
            if basket is not None:
                basket.put(seg_dict)
            elif mode == "reference":  # engines/train_sam_real.py:231-234
                for k, v in seg_dict.items():
                    ref_basket[k][v[1].cpu().detach().numpy()] = v[0].cpu().detach().numpy()
        loss = out["loss"]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        flat = opt.flatten_grads()
        pre.start(data["coord"], data["offset"])
        opt.step(flat_grad=flat)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if basket is not None:
        basket.flush()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    if basket is not None:
        basket.close()
    return ms


for mode in ("default", "seg_dict_only", "copy_only", "full", "reference", "default"):
    print("%-14s %.3f ms/step" % (mode, run(mode)))
