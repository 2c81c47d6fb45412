"""Does any kernel read memory it (or a predecessor) has not written?  Fill the caching allocator's pool with NaN, free it, and
run training steps: every `torch.empty` the step allocates (activation arena, workspaces, geometry, gradients) then starts
out as NaN, and a "garbage x 0" pattern anywhere turns the loss or a gradient into NaN.  usage: python tools/gpu/poison_check.py [points]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ao_amd.ptv2 as ptv2
from ao_amd import synth
from ao_amd.ptv2 import parallel
from ao_amd.ptv2.optim import FlatAdamW

pts = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
seg = ptv2.DefaultSegmentor(dict(ptv2.S3DIS_BACKBONE)).to(dev).train()
seg.backbone.native_param_grads = "direct"
opt = FlatAdamW(seg.parameters(), lr=0.006, weight_decay=0.05)
b = synth.scene_batch([0], point_max=pts, room=1)
data = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}

def poison():
    blocks = []
    for nbytes in [1 << 12, 1 << 16, 1 << 20, 1 << 24, 1 << 26, 1 << 28, 1 << 30, 1 << 31]:
        for _ in range(6 if nbytes < (1 << 30) else 3):
            blocks.append(torch.full((nbytes // 4,), float("nan"), device=dev))
    torch.cuda.synchronize()
    del blocks  # back to the caching allocator: the next torch.empty of a fitting size gets NaN-filled memory

pre = parallel.GeometryPrefetcher(seg.backbone, dev)
bad = 0
for step in range(4):
    poison()
    pre.start(data["coord"], data["offset"])
    loss = seg(dict(data, geometry=pre.take()))["loss"]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    flat = opt.flatten_grads()
    ok = bool(torch.isfinite(loss)) and bool(torch.isfinite(flat).all())
    print("step %d: loss %.6f grads finite %s" % (step, float(loss), bool(torch.isfinite(flat).all())))
    bad += not ok
    opt.step(flat_grad=flat)
print("POISON CHECK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
