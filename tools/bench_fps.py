#!/usr/bin/env python
"""FPS alone: microseconds per sample at the bench scene (120 k points -> every 4th), checked against the C oracle on a prefix.
usage: tools/bench_fps.py [points] [stride]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ao_amd import pointops, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 4
b = synth.scene_batch([0], point_max=n, room=1)
xyz = torch.from_numpy(b["coord"]).cuda()
off = torch.tensor([xyz.shape[0]], dtype=torch.int32).cuda()
noff = torch.tensor([xyz.shape[0] // stride], dtype=torch.int32).cuda()
idx = pointops.farthest_point_sampling(xyz, off, noff)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2):
    idx = pointops.farthest_point_sampling(xyz, off, noff)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 2
print("fps %d -> %d samples: %.2f ms, %.3f us per sample" % (xyz.shape[0], int(noff[-1]), dt * 1e3, dt * 1e6 / int(noff[-1])))
from oracle import pointops_ref as P  # noqa: E402

m = 600
ref = P.farthest_point_sampling(xyz.cpu(), off.cpu(), torch.tensor([m], dtype=torch.int32))
print("first %d samples equal the oracle's: %s" % (m, bool(torch.equal(idx[:m].cpu(), ref))))
