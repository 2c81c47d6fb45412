#!/usr/bin/env python
"""The scene geometry of the bench scene alone, N times (for a rocprofv3 kernel trace of just the geometry):
tools/geo_only.py [repeats] [cfg: s3dis|scannet]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ao_amd.ptv2 as ptv2
from ao_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = sys.argv[2] if len(sys.argv) > 2 else "s3dis"
dev = torch.device("cuda")
if cfg == "s3dis":
    b = synth.scene_batch([0], point_max=120000, room=1)
    backbone = ptv2.S3DIS_BACKBONE
else:
    b = synth.scene_batch([0, 1], point_max=100000, room=1)
    backbone = ptv2.SCANNET_BACKBONE
data = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
seg = ptv2.DefaultSegmentor(backbone).to(dev).train()
with torch.no_grad():
    for _ in range(3):
        seg.backbone.geometry(data["coord"], data["offset"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        seg.backbone.geometry(data["coord"], data["offset"])
    e1.record()
    torch.cuda.synchronize()
print("geometry %.3f ms per scene batch" % (e0.elapsed_time(e1) / reps))
