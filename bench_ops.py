#!/usr/bin/env python
"""bench_ops.py -- kNN and FPS micro-benchmarks (BASELINE.json: "knn+FPS us/query"), one JSON line each.

  python bench_ops.py [--points 80000] [--reps 20] [--cpu]

kNN: self query k=16 and cross query k=3 (interpolation shape, 1/6.3 of the points as support) on a synthetic
S3DIS-shaped scene; FPS: stride-4 sampling (PTv1 call site).  Times are HIP-event times of the
whole op (grid build + query [+ tie pass] for kNN) divided by the number of queries / samples.  --cpu also times
the C oracle (single thread = literal reference algorithm, and OpenMP over queries) on a bounded subset.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def gpu_time(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=80000)
    ap.add_argument("--scenes", type=int, default=1)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    from ao_amd import pointops, synth

    b = synth.scene_batch(list(range(args.scenes)), point_max=args.points, room=1)
    xyz, off = torch.from_numpy(b["coord"]).cuda(), torch.from_numpy(b["offset"]).cuda()
    n = xyz.shape[0]
    coarse = xyz[::6].contiguous()
    coff = torch.tensor([coarse.shape[0]], dtype=torch.int32, device="cuda") if args.scenes == 1 else None
    out = []
    t = gpu_time(lambda: pointops.knn_query_dist2(16, xyz, off), args.reps)
    out.append(dict(op="knn_query", k=16, m=n, n=n, us_total=t, us_per_query=t / n,
                    algorithmic_GBps=(12 * n + 12 * n + 8 * n * 16) / (t * 1e-6) / 1e9))
    if coff is not None:
        t = gpu_time(lambda: pointops.knn_query_dist2(3, coarse, coff, xyz, off), args.reps)
        out.append(dict(op="knn_query", k=3, m=n, n=coarse.shape[0], us_total=t, us_per_query=t / n))
    noff = (off // 4).int()
    m = int(noff[-1])
    t = gpu_time(lambda: pointops.farthest_point_sampling(xyz, off, noff), max(1, args.reps // 10))
    out.append(dict(op="farthest_point_sampling", n=n, m=m, us_total=t, us_per_sample=t / m))
    if args.cpu:
        from oracle import pointops_ref as P

        sub = 4000
        cx, co = torch.from_numpy(b["coord"]), torch.from_numpy(b["offset"])
        q = cx[:sub].contiguous()
        qo = torch.tensor([sub], dtype=torch.int32)
        for mt in (False, True):
            t0 = time.perf_counter()
            P.knn_query_raw(16, cx[: int(co[0])], co[:1], q, qo, mt=mt)
            dt = time.perf_counter() - t0
            out.append(dict(op="knn_query", impl="cpu oracle" + (" openmp" if mt else " 1 thread"), k=16, m=sub,
                            n=int(co[0]), us_per_query=dt * 1e6 / sub, cores=os.cpu_count() if mt else 1))
        fx = cx[:20000].contiguous()
        t0 = time.perf_counter()
        P.farthest_point_sampling(fx, torch.tensor([20000], dtype=torch.int32), torch.tensor([500], dtype=torch.int32))
        dt = time.perf_counter() - t0
        out.append(dict(op="farthest_point_sampling", impl="cpu oracle 1 thread", n=20000, m=500,
                        us_per_sample=dt * 1e6 / 500))
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
