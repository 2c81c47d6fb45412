import sys, time, torch
sys.path.insert(0, "/root/repo")
import ao_amd.ptv2 as ptv2
from ao_amd import synth
from ao_amd.ptv2.optim import FlatAdamW
torch.manual_seed(0)
cfg = dict(ptv2.SCANNET_BACKBONE)
seg = ptv2.DefaultSegmentor(cfg).cuda().train()
opt = FlatAdamW(seg.parameters(), lr=0.006, weight_decay=0.05)
b = synth.scene_batch([0, 1], point_max=100000, in_channels=9, num_classes=20, room=2)
data = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
losses = []
for it in range(8):
    if it == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = seg(data)["loss"]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    losses.append(float(loss.detach()))
torch.cuda.synchronize()
print("scannet cfg, %d points: %.1f ms/step, losses %s" % (data["coord"].shape[0], 1e3 * (time.perf_counter() - t0) / 5, ["%.3f" % l for l in losses]))
