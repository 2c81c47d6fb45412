#!/usr/bin/env python
"""Per-stage timing of the fused attention at the four S3DIS resolutions (kernel timer of the library):
   python tools/bench_gva_stages.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, pointops, synth
from ao_amd.ptv2.gva import _HipImpl, inverse_table

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
LEVELS = [(120000, 48, 6), (18900, 96, 12), (4500, 192, 24), (1074, 384, 48)]  # stage sizes of the 120 k-point bench scene
for n, c, g in LEVELS:
    k = 16
    pts = synth.room_scene(seed=1, room=1, point_max=n, voxel=0.04 * (120000 / n) ** 0.5)  # coarser levels: coarser voxels
    n = pts.shape[0]
    coord = torch.from_numpy(pts).cuda()
    offset = torch.tensor([n], dtype=torch.int32, device="cuda")
    idx, _ = pointops.knn_query(k, coord, offset)
    inverse_table(idx)
    torch.manual_seed(0)
    dev = "cuda"
    W1 = torch.randn(n, k, g, device=dev, requires_grad=True)
    sc = torch.rand(g, device=dev, requires_grad=True); sh = torch.randn(g, device=dev, requires_grad=True)
    Ww2 = (torch.randn(g, g, device=dev) / g ** 0.5).requires_grad_(True); bw2 = torch.randn(g, device=dev, requires_grad=True)
    v = torch.randn(n, c, device=dev, requires_grad=True)
    a = torch.randn(c, 3, device=dev, requires_grad=True); b = torch.randn(c, device=dev, requires_grad=True)
    kW = torch.randn(n, g, device=dev, requires_grad=True); qW = torch.randn(n, g, device=dev, requires_grad=True)
    M = torch.randn(c, g, device=dev, requires_grad=True); cW = torch.randn(g, device=dev, requires_grad=True)
    for it in range(reps + 2):
        if it == 2:
            torch.cuda.synchronize(); _lib.kernel_timer(True)
        lg = _HipImpl.logits(kW, qW, a, b, M, cW, coord, idx)
        out = _HipImpl.aggregate(W1, sc, sh, Ww2, bw2, v, a, b, coord, idx)
        tot = sum(o.sum() for o in out) + sum(o.float().sum() for o in lg)
        tot.backward()
    torch.cuda.synchronize(); _lib.kernel_timer(False)
    r = _lib.kernel_timer_read()
    print("n=%d c=%d g=%d: " % (n, c, g) + "  ".join("%s %.0fus" % (kk.replace("_kernel", ""), vv["avg_us"]) for kk, vv in
                                                       sorted(r.items(), key=lambda kv: -kv[1]["total_us"])))
