"""Wall time of the step phases (each bracketed by synchronize): geometry / forward / backward / optimizer."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ao_amd.ptv2 as ptv2
from ao_amd import synth

torch.manual_seed(0)
dev = torch.device("cuda")
seg = ptv2.DefaultSegmentor(ptv2.S3DIS_BACKBONE).to(dev).train()
opt = torch.optim.AdamW(seg.parameters(), lr=0.006, weight_decay=0.05, fused=True)
b = synth.scene_batch([0], point_max=int(sys.argv[1]) if len(sys.argv) > 1 else 120000, room=1)
data = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
bb = seg.backbone
def sync(): torch.cuda.synchronize()
acc = dict(geometry=0.0, forward=0.0, backward=0.0, optimizer=0.0)
for it in range(13):
    sync(); t0 = time.perf_counter()
    with torch.no_grad():
        geo = bb.geometry(data["coord"], data["offset"].int())
    sync(); t1 = time.perf_counter()
    logits = bb(data, geometry=geo)
    loss = torch.nn.functional.cross_entropy(logits, data["segment"], ignore_index=-1)
    sync(); t2 = time.perf_counter()
    opt.zero_grad(set_to_none=True); loss.backward()
    sync(); t3 = time.perf_counter()
    opt.step()
    sync(); t4 = time.perf_counter()
    if it >= 3:
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): acc[k] += v / 10
print({k: round(v * 1e3, 2) for k, v in acc.items()}, "ms; total", round(sum(acc.values()) * 1e3, 2))
