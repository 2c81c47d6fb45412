"""Count host-side ops / kernel launches of one training step (torch.profiler), to find launch-count hot spots."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import ao_amd.ptv2 as ptv2
from ao_amd import synth

torch.manual_seed(0)
dev = torch.device("cuda")
seg = ptv2.DefaultSegmentor(ptv2.S3DIS_BACKBONE).to(dev).train()
from ao_amd.ptv2.optim import FlatAdamW
opt = FlatAdamW(seg.parameters(), lr=0.006, weight_decay=0.05)
b = synth.scene_batch([0], point_max=int(sys.argv[1]) if len(sys.argv) > 1 else 120000, room=1)
data = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
def geometry():
    with torch.no_grad():
        return seg.backbone.geometry(data["coord"], data["offset"])
def step(geo):
    loss = seg(dict(data, geometry=geo))["loss"]; opt.zero_grad(set_to_none=True); loss.backward(); opt.step(); return loss
for _ in range(3): step(geometry())
torch.cuda.synchronize()
def census(title, fn):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn(); torch.cuda.synchronize()
    ka = prof.key_averages()
    rows = sorted(ka, key=lambda e: -e.count)
    print("==== %s" % title)
    print("%-70s %6s %10s %10s" % ("op / kernel", "count", "cpu_us", "self_cpu"))
    for e in rows[:45]:
        print("%-70s %6d %10.0f %10.0f" % (e.key[:70], e.count, e.cpu_time_total, e.self_cpu_time_total))
    print("total self cpu ms:", sum(e.self_cpu_time_total for e in ka) / 1e3)
    return out
geo = census("geometry (side stream in the bench)", geometry)
census("forward + loss + backward + optimizer", lambda: step(geo))
