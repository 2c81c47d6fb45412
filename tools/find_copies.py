#!/usr/bin/env python
"""Which python lines issue device-to-device copies / fills / small eager torch kernels in one training step:
tools/find_copies.py  (torch.profiler with stacks; prints op, count, top user frame)"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ao_amd.ptv2 as ptv2
from ao_amd import synth
from ao_amd.ptv2.optim import FlatAdamW

dev = torch.device("cuda")
b = synth.scene_batch([0], point_max=120000)
data = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
seg = ptv2.DefaultSegmentor(dict(ptv2.S3DIS_BACKBONE)).to(dev).train()
opt = FlatAdamW(seg.parameters(), lr=0.006, weight_decay=0.05)


def step():
    loss = seg(data)["loss"]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step(flat_grad=opt.flatten_grads())


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::add_",
                   "aten::mul", "aten::cat", "aten::index", "aten::sqrt", "aten::to", "aten::_to_copy", "aten::empty_like"):
        frame = next((s for s in ev.stack if "/ao_amd/" in s or "bench" in s or "find_copies" in s), ev.stack[0] if ev.stack else "?")
        cnt[(ev.name, frame.split("/ao_amd/")[-1][:90])] += 1
for (name, frame), c in cnt.most_common(45):
    print("%4d  %-18s %s" % (c, name, frame))
