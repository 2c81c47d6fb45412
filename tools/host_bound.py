"""Is the training step host-bound?  Host issue time (no sync) vs GPU-complete time per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ao_amd.ptv2 as ptv2
from ao_amd import synth

pts = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
seg = ptv2.DefaultSegmentor(dict(ptv2.S3DIS_BACKBONE)).to(dev).train()
from ao_amd.ptv2.optim import FlatAdamW
opt = FlatAdamW(seg.parameters(), lr=0.006, weight_decay=0.05)
seg.backbone.native_param_grads = os.environ.get("AO_AMD_PARAM_GRADS", "direct")
b = synth.scene_batch([0], point_max=pts, room=1)
data = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}

from ao_amd.ptv2 import parallel
pre = parallel.GeometryPrefetcher(seg.backbone, dev) if os.environ.get("AO_AMD_PREFETCH", "1") == "1" else None
if pre:
    pre.start(data["coord"], data["offset"])

def step(marks=None):
    t = time.perf_counter()
    batch = data if pre is None else dict(data, geometry=pre.take())
    loss = seg(batch)["loss"]
    t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    if pre:
        pre.start(data["coord"], data["offset"])
    t2 = time.perf_counter()
    opt.step(flat_grad=opt.flatten_grads())
    t3 = time.perf_counter()
    if marks is not None:
        marks.append((t1 - t, t2 - t1, t3 - t2))

for _ in range(5):
    step()
torch.cuda.synchronize()
marks = []
t0 = time.perf_counter()
for _ in range(10):
    step(marks)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
f = sum(m[0] for m in marks) / 10; bw = sum(m[1] for m in marks) / 10; o = sum(m[2] for m in marks) / 10
print("points %d: host issue %.2f ms/step (fwd %.2f bwd %.2f opt %.2f), complete %.2f ms/step"
      % (pts, 1e3 * t_issue / 10, 1e3 * f, 1e3 * bw, 1e3 * o, 1e3 * t_all / 10))
