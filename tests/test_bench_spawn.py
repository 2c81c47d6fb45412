"""bench.py's own multi-rank launcher (reference: pointcept/engines/launch.py:74-135 mp.spawn + init_process_group)
without a GPU: the parent must start N fresh rank processes with the rendezvous environment, relay exactly rank 0's
JSON line, and exit non-zero when a rank fails.  The ranks run bench.py's dry-run leg (rendezvous + timing protocol
over gloo, no model); the real N-rank step runs in tests/test_gpu_bench_spawn.py."""
import json
import os
import subprocess
import sys

from tests.conftest import ROOT


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, timeout=300)


def test_parent_spawns_ranks_and_relays_rank0_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--points", "1000"], AO_AMD_BENCH_DRYRUN="1")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # exactly ONE line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["launcher"] == "bench.py spawn"
    assert out["points_per_step"] == 2000.0  # SUM over ranks


def test_under_an_external_launcher_each_process_is_a_rank():
    # what `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` does: ranks already exist
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    procs = []
    for rank in range(2):
        e = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=port, AO_AMD_BENCH_DRYRUN="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                                       "--warmup", "0"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-800:] for o in outs]
    assert json.loads(outs[0][0].decode())["n_gpus"] == 2
    assert outs[1][0].decode().strip() == ""  # only rank 0 prints


def test_failing_rank_fails_the_job():
    import pytest
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: the ranks would succeed")
    # no GPU in this container: every real rank stops at "bench.py needs a GPU"; the parent must report failure
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert r.stdout.decode().strip() == ""
    assert b"job aborted" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], RANK="0", WORLD_SIZE="2", LOCAL_RANK="0",
             AO_AMD_BENCH_DRYRUN="1")
    assert r.returncode != 0 and b"WORLD_SIZE=2" in r.stderr
