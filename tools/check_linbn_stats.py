#!/usr/bin/env python
"""Batch statistics of a Linear output two ways (GEMM-epilogue tile records vs the shifted one-pass kernel) against float64."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib
import ao_amd.ptv2.block  # noqa: F401

L = _lib.lib()
for n, cin, cout, shift in [(1003, 48, 96, 0.0), (1003, 48, 96, 5.0), (6000, 48, 48, 0.0), (260, 192, 384, 1.0), (120000, 48, 96, 3.0)]:
    torch.manual_seed(0)
    x = torch.randn(n, cin, device="cuda"); w = torch.randn(cout, cin, device="cuda") / cin ** 0.5
    b = torch.randn(cout, device="cuda") * shift
    h64 = x.double() @ w.double().t() + b.double()
    m64, v64 = h64.mean(0), h64.var(0, unbiased=False)
    r64 = (v64 + 1e-5).rsqrt()
    h = torch.empty(n, cout, device="cuda")
    st = torch.zeros(int(L.bn_tiles_floats(n, cout)), device="cuda")
    arr = ctypes.c_void_p * 1
    _lib.check(L.rows_gemm_fused_hip_launcher(n, cout, cin, 1, 0, arr(x.data_ptr()), arr(w.data_ptr()), 0, arr(b.data_ptr()),
                                              arr(h.data_ptr()), 0, None, None, arr(st.data_ptr()), _lib.stream_ptr()), "gemm")
    mean, rstd = torch.empty(cout, device="cuda"), torch.empty(cout, device="cuda")
    g, be = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    _lib.check(L.bn_tiles_finalize_hip_launcher(n, cout, st.data_ptr(), g.data_ptr(), be.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                None, None, None, None, None, 1e-5, 0.1, _lib.stream_ptr()), "fin")
    ws = _lib.workspace(L.dense_workspace_bytes(n, cout, cout), x.device)
    mean2, rstd2 = torch.empty(cout, device="cuda"), torch.empty(cout, device="cuda")
    _lib.check(L.bn_stats_hip_launcher(n, cout, h.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), None, None, None, 1e-5, 0.1,
                                       ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "stats")
    torch.cuda.synchronize()
    f = lambda a, r: float(((a.double() - r) / r.abs().clamp_min(1e-3)).abs().max())
    print("n=%6d %3d->%3d shift %.0f: tiles mean %.1e rstd %.1e | one-pass mean %.1e rstd %.1e | h err %.1e"
          % (n, cin, cout, shift, f(mean, m64), f(rstd, r64), f(mean2, m64), f(rstd2, r64), float((h.double() - h64).abs().max())))
