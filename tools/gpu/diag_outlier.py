"""Diagnostic: single-element outliers between two evaluations of the same step (usage: python diag_outlier.py ENV_A ENV_B)."""
import os, subprocess, sys, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F

def run(envs, out):
    for kv in envs.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1); os.environ[k] = v
    from oracle import ptv2_ref as M
    from ao_amd import synth
    import ao_amd.ptv2 as ptv2
    cfg = dict(M.S3DIS_CFG, drop_path_rate=0.0)
    b = synth.scene_batch([1 + int(os.environ.get("DIAG_SEED", 0)), 2], point_max=int(os.environ.get("DIAG_POINTS", 6000)), in_channels=cfg["in_channels"], num_classes=cfg["num_classes"])
    data = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    m = ptv2.PointTransformerV2(**cfg).cuda(); m.load_state_dict(M.init_state(cfg, seed=17 + int(os.environ.get("DIAG_SEED", 0))), strict=True); m.train()
    logits = m(data); loss = F.cross_entropy(logits, data["segment"], ignore_index=-1)
    g = torch.autograd.grad(loss, list(m.parameters()))
    torch.save({n: x.cpu() for (n, _), x in zip(m.named_parameters(), g)}, out)

if sys.argv[1] == "--child":
    run(sys.argv[2], sys.argv[3]); sys.exit(0)
outs = []
for i, e in enumerate(sys.argv[1:]):
    o = "/tmp/diag_%d.pt" % i
    subprocess.check_call([sys.executable, __file__, "--child", e, o]); outs.append(torch.load(o))
for i in range(1, len(outs)):
    print("== %s vs %s" % (sys.argv[1], sys.argv[1 + i]))
    for n in outs[0]:
        a, b = outs[0][n].flatten().double(), outs[i][n].flatten().double()
        d = (a - b).abs(); med = float(d.median())
        bad = ((d > 30 * max(med, 1e-9)) & (d > 2e-6)).nonzero().flatten().tolist()
        if bad and (os.environ.get("DIAG_ALL") or n.endswith("norm.bias") or n.endswith("norm.weight")):
            print(n, "n=%d med %.2e" % (d.numel(), med), [(j, "%.3e" % float(a[j]), "%.3e" % float(b[j])) for j in bad[:6]])
