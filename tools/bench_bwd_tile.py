"""GPU micro-benchmark: the deep-level attention backward as one launch per tile (gva_bwd_tile.hip) against peb_bwd + the point
kernel, at the bench scene's level sizes.  usage: python tools/bench_bwd_tile.py [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from ao_amd import _lib  # noqa: E402
from tests.test_gpu_gva_tile import _bwd_inputs, _bwd_outputs, _inputs  # noqa: E402
from tools.bench_fwd_tile import timed  # noqa: E402


def main():
    import ao_amd.ptv2.gva  # noqa: F401

    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
    L = _lib.lib()
    st = _lib.stream_ptr()
    shapes = ((19028, 96, 12), (4501, 192, 24), (1074, 384, 48), (1600, 192, 24), (240, 384, 48))
    if "--sweep" in sys.argv:  # how the launch scales with the number of workgroups (rounds of 256 CUs x 1 or 2 resident)
        shapes = tuple((n, 192, 24) for n in (1024, 2048, 3072, 4096, 4501, 6144, 8192)) + tuple((n, 384, 48) for n in (512, 1074, 2048)) + \
            tuple((n, 96, 12) for n in (4096, 8192, 19028))
    for n, c, g in shapes:
        t = _inputs(n, c, g, seed=5)
        k, dev = 16, t["v"].device
        w, g_out, inv_ptr, inv_rows = _bwd_inputs(t, n, c, g)
        o = _bwd_outputs(n, k, c, g, dev)
        gA, g_sw = torch.empty(n, g, c, device=dev), torch.empty(n, g, device=dev)
        ws = _lib.workspace(L.gva_aggregate_workspace_bytes(n, k, c, g), dev)
        p = {key: val.data_ptr() for key, val in t.items() if key != "k"}
        op = {key: val.data_ptr() for key, val in o.items()}

        def staged():
            L.gva_peb_backward_hip_launcher(n, c, g, g_out.data_ptr(), p["Wp2"], p["bp2"], gA.data_ptr(), g_sw.data_ptr(), st)
            L.gva_aggregate_backward_hip_launcher(n, k, c, g, p["W1"], p["sc"], p["sh"], p["Ww2"], p["bw2"], p["v"], p["a"], p["b"],
                                                  p["coord"], p["idx"], w.data_ptr(), g_out.data_ptr(), gA.data_ptr(), g_sw.data_ptr(),
                                                  inv_ptr.data_ptr(), inv_rows.data_ptr(), op["gW1"], op["gsc"], op["gsh"], op["gWw2"],
                                                  op["gbw2"], op["gv"], op["ga"], op["gb"], ws.data_ptr(), ws.numel(), st)

        def fused():
            L.gva_attention_backward_hip_launcher(n, k, c, g, p["W1"], p["sc"], p["sh"], p["Ww2"], p["bw2"], p["v"], p["a"], p["b"],
                                                  p["coord"], p["idx"], w.data_ptr(), g_out.data_ptr(), p["Wp2"], p["bp2"],
                                                  inv_ptr.data_ptr(), inv_rows.data_ptr(), op["gW1"], op["gsc"], op["gsh"], op["gWw2"],
                                                  op["gbw2"], op["gv"], op["ga"], op["gb"], ws.data_ptr(), ws.numel(), st)

        if "--stamps" in sys.argv:  # a -DBT_STAMPS build: phase boundaries of workgroups 0 and nblk / 2 (100 MHz wall clock)
            fused()
            torch.cuda.synchronize()
            tp, pf = 8, 4 * c + 3 * g + g * g
            nblk = (n + tp - 1) // tp
            off = (nblk * pf + 64) * 4
            raw = ws[off:off + 32 * 8].view(torch.int64).cpu().tolist()
            for base in (0, 16):
                t0 = raw[base]
                print("   wg %s stamps (us from start): %s" % ("0" if base == 0 else "mid", " ".join("%.2f" % ((raw[base + k] - t0) / 100.0) for k in range(10))))
        print("n %6d c %3d g %2d: peb_bwd + point + gv + finalize %7.1f us   tile + gv + finalize %7.1f us" %
              (n, c, g, timed(staged, reps), timed(fused, reps)))


if __name__ == "__main__":
    main()
