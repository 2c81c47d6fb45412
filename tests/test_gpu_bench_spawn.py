"""`python bench.py --gpus 2` end to end on a ONE-GPU box: the parent spawns two rank processes, both on cuda:0
(AO_AMD_BENCH_ONE_DEVICE=1), gradients exchanged over gloo (RCCL refuses two ranks on one device), so that the whole
multi-process flow of the bench -- geometry prefetch stream, FlatAdamW.flatten_grads -> FlatGradSync.reduce_flat ->
step(grad_scale), barrier + max-over-ranks timing, rank-0 JSON -- runs once on real hardware.
Reference flow: pointcept/engines/launch.py:74-135, engines/train_sam_pp2s.py:173-200."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_two_ranks_spawned_by_bench_py_on_one_device():
    common = ["--steps", "3", "--warmup", "2", "--points", "20000", "--no-cpu-baseline", "--no-ops", "--no-roofline"]
    two = _bench(["--gpus", "2"] + common, AO_AMD_BENCH_ONE_DEVICE="1", AO_AMD_BENCH_BACKEND="gloo")
    # (Round 3 retried here once on a NaN loss, seen once in 67 runs, always the first process group on a fresh box.  Root cause
    # found in round 4 in the ISA: the "last block arrives" protocol of the in-kernel reductions published its arrival without
    # draining the block's record stores first (gva_common.h: last_block_arrives, dense.hip: bn_finalize_tiles_split_kernel),
    # so the finishing block could read a record slot's previous contents -- on a fresh process, whatever the workspace
    # held.  Fixed there; no retry: a NaN fails the test.)
    assert two["n_gpus"] == 2 and two["config"]["rccl_ranks"] == 2 and two["config"]["launcher"] == "bench.py spawn"
    assert two["config"]["comm_backend"] == "gloo" and two["config"]["grad_sync"] == "flat all-reduce"
    assert two["value"] > 0 and two["config"]["loss"] == two["config"]["loss"]  # finite
    one = _bench(["--gpus", "1"] + common)
    assert one["n_gpus"] == 1 and one["config"]["launcher"] == "single"
    # weak scaling bookkeeping: points per step summed over the ranks
    assert abs(two["value"] * two["ms_per_step"] - 2 * one["value"] * one["ms_per_step"]) <= 0.02 * one["value"] * one["ms_per_step"]
