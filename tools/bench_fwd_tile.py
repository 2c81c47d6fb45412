"""GPU micro-benchmark: the deep-level attention forward as one launch (gva_fwd_tile.hip) against the staged launches, at the
bench scene's level sizes; outputs preallocated, launchers called back to back.  usage: python tools/bench_fwd_tile.py [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from ao_amd import _lib  # noqa: E402
from tests.test_gpu_gva_tile import _inputs  # noqa: E402


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    import ao_amd.ptv2.gva  # noqa: F401

    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
    L = _lib.lib()
    st = _lib.stream_ptr()
    shapes = ((19028, 96, 12), (4501, 192, 24), (1074, 384, 48), (11000, 96, 12), (1600, 192, 24), (240, 384, 48), (30, 512, 64))
    if "--sweep" in sys.argv:  # how the launch scales with the number of workgroups (rounds of the resident slots)
        shapes = tuple((n, 192, 24) for n in (2048, 3072, 4096, 4501, 6144, 8192)) + tuple((n, 384, 48) for n in (512, 1024, 1074, 2048)) + \
            tuple((n, 96, 12) for n in (4096, 8192, 16384, 19028))
    for n, c, g in shapes:
        t = _inputs(n, c, g, seed=5)
        k = 16
        dev = t["v"].device
        out_v, A, sw = torch.empty(n, c, device=dev), torch.empty(n, g, c, device=dev), torch.empty(n, g, device=dev)
        w, out = torch.empty(n, k, g, device=dev), torch.empty(n, c, device=dev)
        p = {key: val.data_ptr() for key, val in t.items() if key != "k"}

        def staged():
            L.gva_aggregate_forward_hip_launcher(n, k, c, g, p["W1"], p["sc"], p["sh"], p["Ww2"], p["bw2"], p["v"], p["a"], p["b"],
                                                 p["coord"], p["idx"], out_v.data_ptr(), A.data_ptr(), sw.data_ptr(), w.data_ptr(), st)
            L.gva_peb_forward_hip_launcher(n, c, g, A.data_ptr(), p["Wp2"], p["bp2"], sw.data_ptr(), out_v.data_ptr(), out.data_ptr(), st)

        def fused(with_a):
            L.gva_attention_forward_hip_launcher(n, k, c, g, p["W1"], p["sc"], p["sh"], p["Ww2"], p["bw2"], p["v"], p["a"], p["b"],
                                                 p["coord"], p["idx"], p["Wp2"], p["bp2"], w.data_ptr(), sw.data_ptr(), out.data_ptr(),
                                                 A.data_ptr() if with_a else 0, st)

        print("n %6d c %3d g %2d: staged %7.1f us   fused+A %7.1f us   fused %7.1f us" %
              (n, c, g, timed(staged, reps), timed(lambda: fused(True), reps), timed(lambda: fused(False), reps)))


if __name__ == "__main__":
    main()
