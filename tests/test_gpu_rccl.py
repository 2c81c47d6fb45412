"""RCCL dry-readiness (VERDICT r2 "do this" #8): the box has ONE MI355X, so the N-GPU exchange cannot run -- but every
statement of it can run once in a 1-rank RCCL group: `init_process_group("nccl", device_id=...)`, the flat all-reduce on
the optimizer's gradient buffer, the two-chunk overlapped form, the barrier / max-over-ranks timing protocol.  What is
left for the 8-GPU node is the transport, not the code path.  Reference: pointcept/engines/launch.py:107-135
(init_process_group("NCCL")), engines/defaults.py:22-43 (DDP)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--points", "30000",
                          "--no-cpu-baseline", "--no-ops", "--no-roofline", *args], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["flat", "flat2"])
def test_flat_all_reduce_runs_on_rccl_in_a_one_rank_group(mode):
    """bench.py with AO_AMD_FORCE_SYNC=1: a 1-rank "nccl" group on cuda:0, the gradient exchange executed every step."""
    base = _bench({"AO_AMD_GRAD_SYNC": "flat"})  # no group at all
    got = _bench({"AO_AMD_FORCE_SYNC": "1", "AO_AMD_GRAD_SYNC": mode, "MASTER_PORT": "29541" if mode == "flat" else "29542"})
    assert got["config"]["comm_backend"] == "nccl" and got["config"]["rccl_ranks"] == 1
    assert got["config"]["grad_sync"].startswith("flat")
    # a sum over one rank changes nothing: the training trajectory is the one of the run without a group, bit for bit
    assert got["config"]["loss"] == base["config"]["loss"], (got["config"]["loss"], base["config"]["loss"])
