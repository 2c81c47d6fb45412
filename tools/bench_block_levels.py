#!/usr/bin/env python
"""Time ONE native Block (forward, backward) at the four S3DIS level shapes of the 120 k-point bench scene, and what
those blocks add up to per step (3 + 3 + 7 + 2 blocks): where the step's Block time sits by level.
usage: tools/bench_block_levels.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ao_amd.ptv2 as ptv2
from ao_amd import pointops, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
b = synth.scene_batch([0], point_max=120000, room=1)
coord = torch.from_numpy(b["coord"]).to(dev)
offset = torch.from_numpy(b["offset"]).to(dev).int()
model = ptv2.PointTransformerV2(**ptv2.S3DIS_BACKBONE).to(dev)
geo = model.geometry(coord, offset)
plan = [(48, 6, 3), (96, 12, 3), (192, 24, 7), (384, 48, 2)]
total = 0.0
for lv, (c, g, nblocks) in zip(geo.levels, plan):
    n = lv.coord.shape[0]
    idx = lv.neighbours(16)
    blk = ptv2.Block(c, g).to(dev).train()
    x = torch.randn(n, c, device=dev).relu_().requires_grad_(True)
    go = torch.randn(n, c, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(reps + 5):
        ev[0].record()
        y = blk([lv.coord, x, lv.offset], idx)[1]
        ev[1].record()
        y.backward(go)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 5:
            tf += ev[0].elapsed_time(ev[1]) / reps
            tb += ev[1].elapsed_time(ev[2]) / reps
        blk.zero_grad(set_to_none=True)
        x.grad = None
    total += nblocks * (tf + tb)
    print("N %6d C %3d G %2d: forward %.3f ms  backward %.3f ms  x %d blocks = %.3f ms/step" % (n, c, g, tf, tb, nblocks, nblocks * (tf + tb)))
print("blocks total %.3f ms/step" % total)
