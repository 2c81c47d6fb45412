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


def test_eight_ranks_on_one_device_under_the_cpu_quota():
    """the only rehearsal of the 8-GPU launch obtainable on a one-GPU box -- eight rank processes (each with its
    launching thread, the HIP runtime's helper threads and a gloo progress thread) on ONE MI355X under the box's cgroup CPU quota
    (16 CPUs on the pool's hosts: 2 per rank).  The ranks must come up, train and agree, nobody may be throttled inside the timed
    region, and every rank reports what it cost the host.  Reference launch: pointcept/engines/launch.py:74-135."""
    out = _bench(["--gpus", "8", "--steps", "5", "--warmup", "3", "--points", "20000", "--no-cpu-baseline", "--no-ops", "--no-roofline",
                  "--no-reference-loop"], AO_AMD_BENCH_ONE_DEVICE="1", AO_AMD_BENCH_BACKEND="gloo")
    assert out["n_gpus"] == 8 and out["config"]["rccl_ranks"] == 8 and out["config"]["comm_backend"] == "gloo"
    assert out["value"] > 0 and out["config"]["loss"] == out["config"]["loss"]  # finite
    host = out["host"]
    ranks = host["per_rank"]
    assert sorted(r["rank"] for r in ranks) == list(range(8))
    for r in ranks:
        assert r["host_cpu_ms"] > 0 and r["process_cpu_ms_per_step"] >= r["host_cpu_ms"] * 0.5 and r["threads"] >= 1, r
    cg = host["cgroup_timed_region"]
    if cg and "throttled_usec" in cg:  # (a box without a cgroup-v2 CPU controller reports nothing)
        # no stall of the ranks: the kernel may COUNT a period as throttled at its very end with no time lost (seen: nr_throttled
        # 1, throttled_usec 0), what matters is the time the cgroup's threads were actually stopped
        assert cg["throttled_usec"] <= 10000, (cg, host["cgroup_cpu_max"], ranks)
    # the ranks' processes together must fit the quota with room to spare: CPU per step summed over the ranks against the
    # quota's CPU time in one step's wall time
    quota = host.get("cgroup_cpu_max")
    if quota and quota.split()[0] != "max":
        cpus = int(quota.split()[0]) / int(quota.split()[1])
        used = sum(r["process_cpu_ms_per_step"] for r in ranks) / out["ms_per_step"]
        assert used < 0.8 * cpus, (used, cpus)
