"""GPU: the A/B switches the library reads ONCE per process (function-local statics), each exercised in a child process: the
Block parity tests (native Block against the op-by-op python Block, forward + every gradient) run once more with the switch set.

  AO_AMD_GEMM_KSPLIT=1   the K-split row GEMM (off by default: profiles/r06_rejected/gemm_ksplit.md) at the widths it covers
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


@pytest.mark.parametrize("env", [{"AO_AMD_SKINNY_BN": "0"}, {"AO_AMD_BWD_POINT": "1"},
                                 {"AO_AMD_TILE_KEEP_A": "1"}, {"AO_AMD_BT_MIXED": "0"}], ids=lambda e: "-".join("%s=%s" % kv for kv in e.items()))
def test_block_parity_with_a_switch_set(env):
    _run(env, "test_native_block_matches_python_block", 12)
