"""Where does the HOST time of a training step go?  cProfile over 20 steps (no GPU synchronisation inside the loop).
usage: python tools/gpu/host_profile.py [points]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ao_amd.ptv2 as ptv2
from ao_amd import synth
from ao_amd.ptv2 import parallel
from ao_amd.ptv2.optim import FlatAdamW

pts = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
seg = ptv2.DefaultSegmentor(dict(ptv2.S3DIS_BACKBONE)).to(dev).train()
seg.backbone.native_param_grads = "direct"
opt = FlatAdamW(seg.parameters(), lr=0.006, weight_decay=0.05)
b = synth.scene_batch([0], point_max=pts, room=1)
data = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
pre = parallel.GeometryPrefetcher(seg.backbone, dev)
pre.start(data["coord"], data["offset"])

def step():
    loss = seg(dict(data, geometry=pre.take()))["loss"]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    flat = opt.flatten_grads()
    pre.start(data["coord"], data["offset"])
    opt.step(flat_grad=flat)

for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(20):
    step()
pr.disable()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print("host issue %.2f ms/step (under cProfile), complete %.2f" % (1e3 * t_issue / 20, 1e3 * (time.perf_counter() - t0) / 20))
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
