"""CPU: the C-ABI library builds/loads and exports every symbol include/ptv2_hip.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ptv2_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b([a-z_0-9]+(?:_launcher|_workspace_bytes|_saved_bytes|_arena_bytes|_tiles_floats|_param_layout|_version|_info|_enable|_select|_stride|_is_on|_kernel_count|_read|_host|_struct_bytes|_precision|_count_pairs|_stamp_us|_graph_mode|_graph_stats|_graph_reset|_defer_mode))\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from ao_amd import _lib
    import ao_amd.ptv2.gva  # noqa: F401  (registers the fused-attention entry points)
    import ao_amd.ptv2.block  # noqa: F401  (registers the block runtime entry points)
    import ao_amd.ptv2.native_model  # noqa: F401  (the model runtime; also checks the struct mirrors' sizes)

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 18, names
    handle = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), "missing export: " + n
    missing = [n for n in names if n not in lib._SIGNATURES]
    assert not missing, "declared in the header but not bound in ao_amd/_lib.py: %s" % missing
    extra = [n for n in lib._SIGNATURES if n not in names]
    assert not extra, "bound but not declared in include/ptv2_hip.h: %s" % extra


def test_host_only_entry_points(lib):
    L = lib.lib()
    assert L.ptv2_abi_version() == lib.EXPECTED_ABI == 11
    assert b"gfx950" in L.ptv2_build_info()
    a = L.knn_query_hip_workspace_bytes(80000, 80000, 1)
    b = L.knn_query_hip_workspace_bytes(240000, 240000, 3)
    assert 0 < a < b < 64 << 20
    assert L.knn_query_hip_workspace_bytes(10, 10, 0) == 0  # invalid b


def test_ops_fail_loudly_without_gpu(lib):
    import torch

    from ao_amd import pointops

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    xyz = torch.zeros(8, 3)
    off = torch.tensor([8], dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pointops.knn_query(2, xyz, off)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pointops.grouping(torch.zeros(8, 2, dtype=torch.int32), torch.zeros(8, 4), xyz)


def test_binding_arity_matches_header(lib):
    """ctypes argtypes must have exactly one entry per parameter declared in the header."""
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    txt = re.sub(r"typedef struct.*?\}\s*\w+\s*;", "", txt, flags=re.S)
    for name, params in re.findall(r"\b([a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", txt):
        if name not in lib._SIGNATURES:
            continue
        params = params.strip()
        count = 0 if params in ("", "void") else params.count(",") + 1
        assert count == len(lib._SIGNATURES[name][1]), (name, count, len(lib._SIGNATURES[name][1]))


def test_model_runtime_rejects_an_invalid_description(lib):
    """The whole-model launchers validate their struct before anything is enqueued (no GPU needed to see it): a zeroed
    ptv2_model has no stages -> saved/workspace size 0 and PTV2_ERR_ARG; the ctypes mirrors have the library's sizeof."""
    import ctypes

    from ao_amd.ptv2 import block, native_model

    L = lib.lib()
    assert L.ptv2_struct_bytes(0) == ctypes.sizeof(block._Blk)
    assert L.ptv2_struct_bytes(1) == ctypes.sizeof(block._BlkGrads)
    assert L.ptv2_struct_bytes(2) == ctypes.sizeof(native_model._Model)
    from ao_amd.ptv2 import gva
    assert L.ptv2_struct_bytes(3) == ctypes.sizeof(gva._BlockArgs)
    assert L.ptv2_struct_bytes(99) == -1
    M = native_model._Model()
    assert L.ptv2_model_saved_bytes(ctypes.addressof(M)) == 0
    assert L.ptv2_model_workspace_bytes(ctypes.addressof(M)) == 0
    assert L.ptv2_model_forward_hip_launcher(ctypes.addressof(M), None, 0, None) == 1   # PTV2_ERR_ARG
    assert L.ptv2_model_backward_hip_launcher(ctypes.addressof(M), None, None, 0, None) == 1
    # matmul precision switch of the calling thread: set / query / restore
    prev = L.ptv2_matmul_precision(-1)
    assert L.ptv2_matmul_precision(1) == prev and L.ptv2_matmul_precision(-1) == 1
    L.ptv2_matmul_precision(prev)
    assert L.ptv2_matmul_precision(-1) == prev


def test_basket_scatter_host_entry_point(lib):
    """basket_scatter_rows_host is plain host code: rows in order, last write wins, out-of-range ids rejected untouched."""
    import numpy as np

    L = lib.lib()
    dst = np.full((6, 3), -100.0, np.float32)
    ids = np.array([4, 1, 4], np.int64)
    src = np.arange(9, dtype=np.float32).reshape(3, 3)
    assert L.basket_scatter_rows_host(dst.ctypes.data, 6, ids.ctypes.data, src.ctypes.data, 3, 3) == 0
    assert dst[1].tolist() == [3, 4, 5] and dst[4].tolist() == [6, 7, 8] and (dst[[0, 2, 3, 5]] == -100).all()
    bad = np.array([0, 6], np.int64)
    before = dst.copy()
    assert L.basket_scatter_rows_host(dst.ctypes.data, 6, bad.ctypes.data, src.ctypes.data, 2, 3) == 1
    assert np.array_equal(dst, before)
