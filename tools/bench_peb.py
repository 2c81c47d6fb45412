#!/usr/bin/env python
"""Grouped positional-bias projection (peb) forward / backward at the four stage shapes: tools/bench_peb.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib
import ao_amd.ptv2.gva  # noqa: F401  (registers the entry points)

L = _lib.lib()
for n, c, g in [(120000, 48, 6), (18900, 96, 12), (4500, 192, 24), (1074, 384, 48), (240, 512, 64)]:
    A = torch.randn(n, g, c, device="cuda"); Wp2 = torch.randn(c, c, device="cuda") / c ** 0.5; bp2 = torch.randn(c, device="cuda")
    sw = torch.rand(n, g, device="cuda"); out_v = torch.randn(n, c, device="cuda"); out = torch.empty(n, c, device="cuda")
    go = torch.randn(n, c, device="cuda"); gA = torch.empty(n, g, c, device="cuda"); gsw = torch.empty(n, g, device="cuda")
    for it in range(8):
        if it == 3:
            torch.cuda.synchronize(); _lib.kernel_timer(True)
        _lib.check(L.gva_peb_forward_hip_launcher(n, c, g, A.data_ptr(), Wp2.data_ptr(), bp2.data_ptr(), sw.data_ptr(), out_v.data_ptr(),
                                                  out.data_ptr(), _lib.stream_ptr()), "peb fwd")
        _lib.check(L.gva_peb_backward_hip_launcher(n, c, g, go.data_ptr(), Wp2.data_ptr(), bp2.data_ptr(), gA.data_ptr(), gsw.data_ptr(),
                                                   _lib.stream_ptr()), "peb bwd")
    torch.cuda.synchronize(); _lib.kernel_timer(False)
    r = _lib.kernel_timer_read()
    ref = out_v + torch.einsum("ngc,gic->ngi", A, Wp2.view(g, c // g, c)).reshape(n, c) + bp2 * sw.repeat_interleave(c // g, 1)
    err = float((out - ref).abs().max())
    print("n=%6d c=%3d g=%2d: " % (n, c, g) + "  ".join("%s %.1fus (%.0f GB/s)" % (k.replace("_kernel", ""), v["avg_us"], v["bytes_per_launch"] / v["avg_us"] / 1e3)
                                                       for k, v in r.items()) + "  max err %.1e" % err)
