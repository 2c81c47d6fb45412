#!/usr/bin/env python
"""Timing of the dense-layer kernels at the four S3DIS resolutions (library kernel timer): tools/bench_dense.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib
from ao_amd.ptv2.block import rows_gemm
from ao_amd.ptv2.layers import RowBatchNorm1d

L = _lib.lib()
for n, c in [(120000, 48), (30000, 96), (7500, 192), (1900, 384)]:
    x = torch.randn(n, c, device="cuda"); gy = torch.randn(n, c, device="cuda"); w = torch.randn(c, c, device="cuda")
    dW = torch.empty(c, c, device="cuda"); db = torch.empty(c, device="cuda")
    ws = _lib.workspace(L.dense_workspace_bytes(n, c, c), x.device)
    bn = RowBatchNorm1d(c).cuda().train()
    for it in range(7):
        if it == 2:
            torch.cuda.synchronize(); _lib.kernel_timer(True)
        rows_gemm(x, w); rows_gemm(gy, w, w_kmajor=True)
        _lib.check(L.linear_wgrad_hip_launcher(n, c, c, gy.data_ptr(), x.data_ptr(), dW.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                               ws.numel(), _lib.stream_ptr()), "wgrad")
        xx = x.clone().requires_grad_(True)
        y = bn(xx, relu=True); y.backward(gy)
    torch.cuda.synchronize(); _lib.kernel_timer(False)
    r = _lib.kernel_timer_read()
    print("n=%d c=%d: " % (n, c) + "  ".join("%s %.1fus (%.0f GB/s)" % (k.replace("_kernel", ""), v["avg_us"], v["bytes_per_launch"] / v["avg_us"] / 1e3)
                                               for k, v in sorted(r.items(), key=lambda kv: -kv[1]["total_us"])))
