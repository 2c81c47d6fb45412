import os, sys, ctypes, torch
sys.path.insert(0, "/root/repo")
from ao_amd import _lib
import ao_amd.ptv2.block
L = _lib.lib()
def arr(ts): return (ctypes.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for n, c in [(120000, 48), (30000, 96), (7500, 192), (1900, 384)]:
    x = torch.randn(n, c, device="cuda"); w = torch.randn(c, c, device="cuda"); y = torch.empty(n, c, device="cuda")
    sc = torch.rand(c, device="cuda"); sh = torch.randn(c, device="cuda"); gy = torch.randn(n, c, device="cuda")
    st = torch.empty(L.bn_tiles_floats(n, c), device="cuda"); dW = torch.empty(c, c, device="cuda")
    mean = torch.empty(c, device="cuda"); rstd = torch.empty(c, device="cuda"); g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
    ws = _lib.workspace(L.dense_workspace_bytes(n, 3 * c, c), x.device)
    s = _lib.stream_ptr()
    f = lambda xsc, xsh, sts: L.rows_gemm_fused_hip_launcher(n, c, c, 1, 0, arr([x]), arr([w]), 0, None, arr([y]), 0, xsc, xsh, sts, s)
    t0 = timeit(lambda: f(None, None, None))
    t1 = timeit(lambda: f(None, None, arr([st])))
    t2 = timeit(lambda: f(sc.data_ptr(), sh.data_ptr(), None))
    t3 = timeit(lambda: L.bn_tiles_finalize_hip_launcher(n, c, st.data_ptr(), g.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None, None, 1e-5, 0.1, s))
    wg = lambda xs_, xh_: L.linear_wgrad_multi_hip_launcher(n, c, c, 1, arr([gy]), arr([x]), arr([dW]), None, xs_, xh_, ws.data_ptr(), ws.numel(), s)
    t4 = timeit(lambda: wg(None, None)); t5 = timeit(lambda: wg(arr([sc]), arr([sh])))
    print("n=%d c=%d: gemm %.1f  +stats %.1f  +xform %.1f | tiles_finalize %.1f | wgrad %.1f  +xform %.1f us" % (n, c, t0, t1, t2, t3, t4, t5))
