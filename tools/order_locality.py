import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from ao_amd import _lib, pointops, synth
from ao_amd.ptv2.gva import _HipImpl, inverse_table

def morton(coord, bits=10):
    c = coord - coord.min(0)
    q = np.minimum((c / (c.max() + 1e-9) * (2 ** bits - 1)).astype(np.uint64), 2 ** bits - 1)
    def spread(x):
        x = x & 0x3FF
        x = (x | (x << 16)) & 0x30000FF
        x = (x | (x << 8)) & 0x300F00F
        x = (x | (x << 4)) & 0x30C30C3
        x = (x | (x << 2)) & 0x9249249
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)

n, c, g, k = 120000, 48, 6, 16
pts = synth.room_scene(seed=1, room=1, point_max=n)
for tag, order in (("as generated", np.arange(len(pts))), ("morton", np.argsort(morton(pts), kind="stable")), ("random", np.random.default_rng(0).permutation(len(pts)))):
    p = np.ascontiguousarray(pts[order])
    coord = torch.from_numpy(p).cuda(); nn = coord.shape[0]
    offset = torch.tensor([nn], dtype=torch.int32, device="cuda")
    idx, _ = pointops.knn_query(k, coord, offset)
    inverse_table(idx)
    torch.manual_seed(0)
    dev = "cuda"
    W1 = torch.randn(nn, k, g, device=dev, requires_grad=True)
    sc = torch.rand(g, device=dev, requires_grad=True); sh = torch.randn(g, device=dev, requires_grad=True)
    Ww2 = (torch.randn(g, g, device=dev) / g ** 0.5).requires_grad_(True); bw2 = torch.randn(g, device=dev, requires_grad=True)
    v = torch.randn(nn, c, device=dev, requires_grad=True)
    a = torch.randn(c, 3, device=dev, requires_grad=True); b = torch.randn(c, device=dev, requires_grad=True)
    kW = torch.randn(nn, g, device=dev, requires_grad=True); qW = torch.randn(nn, g, device=dev, requires_grad=True)
    M = torch.randn(c, g, device=dev, requires_grad=True); cW = torch.randn(g, device=dev, requires_grad=True)
    for it in range(7):
        if it == 2:
            torch.cuda.synchronize(); _lib.kernel_timer(True)
        lg = _HipImpl.logits(kW, qW, a, b, M, cW, coord, idx)
        out = _HipImpl.aggregate(W1, sc, sh, Ww2, bw2, v, a, b, coord, idx)
        (sum(o.sum() for o in out) + sum(o.float().sum() for o in lg)).backward()
        pointops.knn_query(k, coord, offset)
    torch.cuda.synchronize(); _lib.kernel_timer(False)
    r = _lib.kernel_timer_read()
    print(tag + ": " + "  ".join("%s %.0f" % (kk.replace("_kernel", "")[:24], vv["avg_us"]) for kk, vv in sorted(r.items(), key=lambda kv: -kv[1]["total_us"])))
