import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _cpu_quota():
    """CPUs this process may use: the cgroup v2 quota when there is one (the pool's GPU boxes show 256 CPUs under a
    16-CPU quota; one torch thread per visible CPU is throttled to a crawl there), else the affinity mask."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()
        if a != "max":
            n = min(n, max(1, int(a) // int(b)))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch

    torch.set_num_threads(max(1, min(torch.get_num_threads(), _cpu_quota())))  # the CPU oracle's threads


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import pointops_ref

    pointops_ref.build()
