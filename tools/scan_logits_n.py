#!/usr/bin/env python
"""logits forward stage time against the point count at fixed (C, G): separates the size-independent part of the
kernel (prologue, launch, final reduction) from its per-row work: tools/scan_logits_n.py [c g]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, pointops, synth
from ao_amd.ptv2.gva import _HipImpl

c, g = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (192, 24)
k = 16
for n in (64, 256, 1024, 2048, 4500, 9000, 18000, 36000):
    pts = synth.room_scene(seed=1, room=1, point_max=n, voxel=0.04 * (120000 / max(n, 2000)) ** 0.5)[:n]
    n = pts.shape[0]
    coord = torch.from_numpy(pts).cuda()
    offset = torch.tensor([n], dtype=torch.int32, device="cuda")
    idx, _ = pointops.knn_query(k, coord, offset)
    a = torch.randn(c, 3, device="cuda"); b = torch.randn(c, device="cuda")
    kW = torch.randn(n, g, device="cuda"); qW = torch.randn(n, g, device="cuda")
    M = torch.randn(c, g, device="cuda"); cW = torch.randn(g, device="cuda")
    for it in range(8):
        if it == 3:
            torch.cuda.synchronize(); _lib.kernel_timer(True)
        with torch.no_grad():
            _HipImpl.logits(kW, qW, a, b, M, cW, coord, idx)
    torch.cuda.synchronize(); _lib.kernel_timer(False)
    r = _lib.kernel_timer_read()
    print("n=%6d rows=%7d: " % (n, n * k) + "  ".join("%s %.1fus" % (kk.replace("_kernel", ""), vv["avg_us"]) for kk, vv in r.items()))
