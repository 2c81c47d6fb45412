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
opt = torch.optim.AdamW(seg.parameters(), lr=0.006, weight_decay=0.05, fused=True)
b = synth.scene_batch([0], point_max=int(sys.argv[1]) if len(sys.argv) > 1 else 120000, room=1)
data = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
def step():
    loss = seg(data)["loss"]; opt.zero_grad(set_to_none=True); loss.backward(); opt.step(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print("%-60s %6s %10s %10s" % ("op", "count", "cpu_us", "self_cpu"))
for e in rows[:60]:
    print("%-60s %6d %10.0f %10.0f" % (e.key[:60], e.count, e.cpu_time_total, e.self_cpu_time_total))
print("total self cpu ms:", sum(e.self_cpu_time_total for e in ka) / 1e3)
