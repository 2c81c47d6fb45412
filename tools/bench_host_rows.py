#!/usr/bin/env python
"""Timing of the data-side rows (GridSample, SphereCrop, validation counts) on the GPU next to the numpy
restatement of the reference transform on the host: tools/bench_host_rows.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import synth
from ao_amd.ptv2.transform import GridSample, SphereCrop
from ao_amd.ptv2.evaluate import confusion_counts
from oracle import host_ref as H  # checker / CPU baseline only


def gpu_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def cpu_ms(fn, reps=2):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e3


rng = np.random.default_rng(0)
base = synth.room_scene(seed=1, room=2, point_max=400000, density=8000.0, voxel=0.02)
pts = np.concatenate([base + rng.normal(0, 0.01, base.shape).astype(np.float32) for _ in range(4)])
dev = torch.from_numpy(pts).cuda()
gen = torch.Generator(device="cuda").manual_seed(0)
gs = GridSample(grid_size=0.04, keys=("coord",))
n = pts.shape[0]
a = gpu_ms(lambda: gs(dict(coord=dev), generator=gen))
b = cpu_ms(lambda: H.grid_sample_train(pts, 0.04))
kept = gs(dict(coord=dev), generator=gen)["coord"]
print("GridSample 0.04 m: %d -> %d points: GPU %.2f ms (%.0f M pts/s), numpy %.0f ms (x%.0f)" % (n, kept.shape[0], a, n / a / 1e3, b, b / a))
sc = SphereCrop(point_max=80000)
a = gpu_ms(lambda: sc(dict(coord=kept), generator=gen))
kc = kept.cpu().numpy()
b = cpu_ms(lambda: H.sphere_crop(kc, 80000, 1234))
print("SphereCrop %d -> 80000: GPU %.2f ms, numpy %.1f ms (x%.0f)" % (kept.shape[0], a, b, b / a))
m = 4_000_000
t = torch.randint(-1, 13, (m,), device="cuda"); p = torch.randint(0, 13, (m,), device="cuda")
nn = torch.randint(0, m, (m,), device="cuda", dtype=torch.int32)
a = gpu_ms(lambda: confusion_counts(p, t, 13, -1, nn))
tc, pc = t.cpu().numpy(), p.cpu().numpy()[nn.cpu().numpy()]
b = cpu_ms(lambda: H.intersection_and_union(pc, tc, 13, -1))
print("validation counts over %d labels (through the k=1 table): GPU %.3f ms (%.0f GB/s), numpy %.0f ms" % (m, a, m * 28 / a / 1e6, b))
