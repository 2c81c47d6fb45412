"""GPU: cProfile of the reference trainer's run_step statements over this build (launching thread + autograd thread are both
python here: the engine thread's frames appear under loss.backward's native call only as wall time).
usage: python tools/ref_loop_profile.py [steps]"""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, ".")
import ao_amd.ptv2 as ptv2  # noqa: E402
from ao_amd import synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda", 0)
    b = synth.scene_batch([0], point_max=120000)
    host = {k: torch.from_numpy(v).pin_memory() for k, v in b.items()}
    model = ptv2.DefaultSegmentor(backbone=dict(ptv2.S3DIS_BACKBONE, type="PT-v2m2")).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=0.006, weight_decay=0.05)

    def step():
        d = {k: v.cuda(non_blocking=True) for k, v in host.items()}
        loss = model(d)["loss"]
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
