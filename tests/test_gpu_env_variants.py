"""GPU: the A/B switches the library reads ONCE per process (function-local statics), each exercised in a child process: the
Block parity tests (native Block against the op-by-op python Block, forward + every gradient) run once more with the switch set.

  AO_AMD_GEMM_KSPLIT=1   the K-split row GEMM (off by default: profiles/r06_rejected/gemm_ksplit.md) at the widths it covers
  AO_AMD_GV_MERGE=0      grad v and the logits stage's gather as two launches (default: one walk of the inverse lists at n >= 32768)
  AO_AMD_SKINNY_BN=0     skinny input gradients and the q / k BatchNorm reduce as two launches
  AO_AMD_BWD_POINT=1     deep-level attention backward as peb_bwd + point kernel instead of the tile kernel
  AO_AMD_TILE_KEEP_A=1   the forward tile kernel also writes A; the strided weight gradient reads it
  AO_AMD_BT_MIXED=0      the backward tile kernel's last round in 8-point tiles (default: the remainder in 4-point tiles)
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env, select, expect):
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_block.py"), "-q", "-x", "-k", select,
                        "-p", "no:cacheprovider"], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "%d passed" % expect in r.stdout, r.stdout[-500:]


def test_block_parity_with_the_k_split_gemm_on():
    _run({"AO_AMD_GEMM_KSPLIT": "1"}, "test_native_block_matches_python_block and (96-12 or 192-24 or 384-48)", 6)


@pytest.mark.parametrize("env", [{"AO_AMD_GV_MERGE": "0"}, {"AO_AMD_SKINNY_BN": "0"}, {"AO_AMD_BWD_POINT": "1"},
                                 {"AO_AMD_TILE_KEEP_A": "1"}, {"AO_AMD_BT_MIXED": "0"}], ids=lambda e: "-".join("%s=%s" % kv for kv in e.items()))
def test_block_parity_with_a_switch_set(env):
    _run(env, "test_native_block_matches_python_block", 12)


def test_the_merged_grad_v_launch_is_taken_at_the_full_resolution_size():
    """n >= 32768 rows: the merged launch (gva_block.hip); one Block forward + backward at 40 000 points against the two-launch
    form in a second child -- the same bits (the sums run in list order either way)."""
    code = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
from ao_amd import pointops, synth
from ao_amd.ptv2.model import Block
n, c, g, k = 40000, 48, 6, 16
coord = torch.from_numpy(synth.room_cloud(n, seed=3)).cuda()
offset = torch.tensor([n], dtype=torch.int32, device="cuda")
idx, _ = pointops.knn_query(k, coord, offset)
torch.manual_seed(0)
blk = Block(c, g, drop_path_rate=0.0).cuda().train()
x = torch.randn(n, c, device="cuda").relu_().requires_grad_(True)
y = blk([coord, x, offset], idx)[1]
grads = torch.autograd.grad(y, [x] + list(blk.parameters()), torch.randn(n, c, device="cuda"))
torch.save([t.cpu() for t in (y.detach(),) + tuple(grads)], sys.argv[1])
''' % ROOT
    import tempfile

    import torch

    with tempfile.TemporaryDirectory() as d:
        outs = []
        for tag, env in (("merged", {}), ("two", {"AO_AMD_GV_MERGE": "0"})):
            path = os.path.join(d, tag + ".pt")
            r = subprocess.run([sys.executable, "-c", code, path], cwd=ROOT, env=dict(os.environ, **env), capture_output=True,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(torch.load(path))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
