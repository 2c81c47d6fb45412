"""GPU: where the launching thread's time goes in the reference trainer's run_step statements over this build (bench.py's
`reference_loop` leg, AdamW variant): wall time of every statement, averaged over the steps, no synchronisation inside the loop.
usage: python tools/ref_loop_breakdown.py [steps] [AdamW|fused|FlatAdamW] [direct]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import ao_amd.ptv2 as ptv2  # noqa: E402
from ao_amd import synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    kind = sys.argv[2] if len(sys.argv) > 2 else "AdamW"
    direct = "direct" in sys.argv[3:]
    dev = torch.device("cuda", 0)
    cfg = dict(ptv2.S3DIS_BACKBONE)
    if direct:
        cfg["native_param_grads"] = "direct"
    b = synth.scene_batch([0], point_max=120000)
    host = {k: torch.from_numpy(v).pin_memory() for k, v in b.items()}
    model = ptv2.DefaultSegmentor(backbone=dict(cfg, type="PT-v2m2")).to(dev).train()
    if kind == "FlatAdamW":
        from ao_amd.ptv2.optim import FlatAdamW

        opt = FlatAdamW(params=model.parameters(), lr=0.006, weight_decay=0.05)
    elif kind == "fused":
        opt = torch.optim.AdamW(model.parameters(), lr=0.006, weight_decay=0.05, fused=True)
    else:  # (fused / foreach left at None: torch picks the foreach implementation for CUDA parameters)
        opt = torch.optim.AdamW(model.parameters(), lr=0.006, weight_decay=0.05)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10 ** 6], gamma=0.1)
    names = ["to_device", "forward", "zero_grad", "backward", "optimizer.step", "scheduler.step"]
    acc = dict.fromkeys(names, 0.0)

    def step(record):
        t = [time.perf_counter()]
        d = {k: v.cuda(non_blocking=True) for k, v in host.items()}
        t.append(time.perf_counter())
        loss = model(d)["loss"]
        t.append(time.perf_counter())
        opt.zero_grad()
        t.append(time.perf_counter())
        loss.backward()
        t.append(time.perf_counter())
        opt.step()
        t.append(time.perf_counter())
        sched.step()
        t.append(time.perf_counter())
        if record:
            for i, nm in enumerate(names):
                acc[nm] += t[i + 1] - t[i]

    for _ in range(8):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print("%s%s: %.2f ms/step (host issue %.2f ms/step)" % (kind, " direct" if direct else "", 1e3 * total / steps, 1e3 * issue / steps))
    for nm in names:
        print("  %-16s %6.2f ms" % (nm, 1e3 * acc[nm] / steps))


if __name__ == "__main__":
    main()
