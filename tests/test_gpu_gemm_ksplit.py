"""GPU: the K-split row GEMM (ao_amd/csrc/gemm.hip rows_gemm_ksplit_kernel) is off by default (measured slower,
profiles/r06_rejected/gemm_ksplit.md) and is kept correct behind AO_AMD_GEMM_KSPLIT=1: the Block parity tests at the three
widths it covers run once more in a child process with the switch on (the library reads it once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_parity_with_the_k_split_gemm_on():
    env = dict(os.environ, AO_AMD_GEMM_KSPLIT="1")
    sel = "test_native_block_matches_python_block and (96-12 or 192-24 or 384-48)"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_block.py"), "-q", "-x", "-k", sel,
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "6 passed" in r.stdout, r.stdout[-500:]
