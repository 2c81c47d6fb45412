"""GPU: the deferred parameter-gradient sums ("riders", ao_amd/csrc/gva_common.h / abi.hip) -- finalizes that run on trailing
workgroups of an independent later launch -- against the same build with every finalize as a launch of its own
(AO_AMD_RIDERS=0, read once per process: two child processes).  Every parameter gradient of one training step must agree
(the two forms add the same records in different, each fixed, orders)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, %(root)r)
from oracle import ptv2_ref as M
from ao_amd import synth
import ao_amd.ptv2 as ptv2
cfg = dict(M.S3DIS_CFG if %(tag)r == "s3dis" else M.SCANNET_CFG, drop_path_rate=0.0)
b = synth.scene_batch([3, 4], point_max=%(points)d, in_channels=cfg["in_channels"], num_classes=cfg["num_classes"])
data = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
m = ptv2.PointTransformerV2(**cfg).cuda()
m.load_state_dict(M.init_state(cfg, seed=5), strict=True)
m.train()
logits = m(data)
loss = F.cross_entropy(logits, data["segment"], ignore_index=-1)
grads = torch.autograd.grad(loss, list(m.parameters()))
np.savez(%(out)r, loss=float(loss), **{"g%%d" %% i: g.cpu().numpy() for i, g in enumerate(grads)})
"""


@pytest.mark.parametrize("tag,points", [("s3dis", 20000), ("scannet", 9000)])
def test_riders_leave_every_parameter_gradient_unchanged(tmp_path, tag, points):
    out = {}
    for mode in ("1", "0"):
        path = str(tmp_path / ("grads_%s.npz" % mode))
        env = dict(os.environ, AO_AMD_RIDERS=mode, AO_AMD_MODEL="native")
        code = CHILD % dict(root=ROOT, tag=tag, points=points, out=path)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[mode] = np.load(path)
    a, b = out["1"], out["0"]
    assert abs(float(a["loss"]) - float(b["loss"])) == 0.0  # the forward has no riders
    worst = 0.0
    for k in a.files:
        if k == "loss":
            continue
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        assert np.isfinite(x).all()
        worst = max(worst, float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30)))
    assert worst < 2e-6, worst
