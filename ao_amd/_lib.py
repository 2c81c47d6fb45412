"""ctypes binding of ao_amd/lib/libptv2_hip.so (C ABI: include/ptv2_hip.h).

There is deliberately NO fallback: if the HIP library is missing, fails to load,
or a launcher returns a non-zero status, a RuntimeError is raised.  Nothing in
this package computes on the CPU or through the oracle.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libptv2_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_c_int, _c_size, _vp = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); pointers are passed as integers (tensor.data_ptr())
_SIGNATURES = {
    "ptv2_abi_version": (_c_int, []),
    "ptv2_build_info": (ctypes.c_char_p, []),
    "ptv2_struct_bytes": (ctypes.c_longlong, [_c_int]),
    "ptv2_matmul_precision": (_c_int, [_c_int]),
    "ptv2_profile_enable": (_c_int, [_c_int]),
    "ptv2_profile_select": (_c_int, [_c_int]),
    "ptv2_profile_stride": (_c_int, [_c_int]),
    "ptv2_profile_is_on": (_c_int, []),
    "ptv2_profile_kernel_count": (_c_int, []),
    "ptv2_profile_read": (_c_int, [_c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double),
                                   ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double)]),
    "ptv2_profile_empty_stamp_us": (ctypes.c_double, [_vp, _c_int]),
    "ptv2_graph_mode": (_c_int, [_c_int]),
    "ptv2_wgrad_defer_mode": (_c_int, [_c_int]),
    "ptv2_graph_stats": (_c_int, [ctypes.POINTER(ctypes.c_double), _c_int]),
    "ptv2_graph_reset": (_c_int, []),
    "knn_query_hip_workspace_bytes": (_c_size, [_c_int] * 3),
    "knn_query_hip_launcher": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _c_size, _vp]),
    "knn_query_grid_hip_launcher": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int,
                                             _vp, _c_size, _vp]),
    "knn_query_count_pairs": (_c_int, [_vp]),
    "farthest_point_sampling_hip_workspace_bytes": (_c_size, [_c_int] * 2),
    "farthest_point_sampling_hip_launcher": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _vp, _c_size, _vp]),
    "grouping_forward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 4),
    "grouping_backward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 4),
    "interpolation_weights_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 4),
    "interpolation_forward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 5),
    "interpolation_backward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 5),
    "interpolation_backward_gather_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 6),
    "subtraction_forward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 5),
    "subtraction_backward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 5),
    "aggregation_forward_hip_launcher": (_c_int, [_c_int] * 4 + [_vp] * 6),
    "aggregation_backward_hip_launcher": (_c_int, [_c_int] * 4 + [_vp] * 9),
    "attention_relation_step_forward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 7),
    "attention_relation_step_backward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 10),
    "attention_fusion_step_forward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 6),
    "attention_fusion_step_backward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 8),
    "grid_pool_hip_workspace_bytes": (_c_size, [_c_int] * 2),
    "grid_pool_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 2 + [ctypes.c_float] + [_vp] * 6 + [_c_int, _vp, _c_size, _vp]),
    "inverse_table_hip_workspace_bytes": (_c_size, [_c_int] * 2),
    "inverse_table_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 4 + [_c_size, _vp]),
    "inverse_tables_hip_workspace_bytes": (_c_size, [_c_int, _vp]),
    "inverse_tables_hip_launcher": (_c_int, [_c_int, _vp, _vp, _c_size, _vp]),
    "segment_minmax_hip_workspace_bytes": (_c_size, [_c_int]),
    "segment_minmax_hip_launcher": (_c_int, [_c_int] + [_vp] * 5 + [_c_size, _vp]),
    "pool_max_forward_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 6),
    "pool_max_backward_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 4),
    "segment_sum_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 5),
    "dense_workspace_bytes": (_c_size, [_c_int] * 3),
    "adamw_flat_hip_launcher": (_c_int, [ctypes.c_longlong] + [_vp] * 4 + [ctypes.c_float] * 5 + [_c_int, ctypes.c_float, _vp]),
    "grid_sample_keys_hip_launcher": (_c_int, [_c_int, _vp] + [ctypes.c_float] * 3 + [_c_int] + [_vp] * 4),
    "center_dist2_hip_launcher": (_c_int, [_c_int] + [_vp] * 4),
    "seg_confusion_hip_launcher": (_c_int, [ctypes.c_longlong, _c_int, _c_int, _vp, ctypes.c_longlong] + [_vp] * 4),
    "basket_scatter_rows_host": (_c_int, [_vp, ctypes.c_longlong, _vp, _vp, ctypes.c_longlong, _c_int]),
    "cross_entropy_workspace_bytes": (_c_size, [_c_int]),
    "cross_entropy_forward_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 2 + [_c_int] + [_vp] * 5 + [_c_size, _vp]),
    "cross_entropy_backward_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 2 + [_c_int] + [_vp] * 5),
    "bn_stats_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 6 + [ctypes.c_float] * 2 + [_vp, _c_size, _vp]),
    "bn_apply_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 5 + [_c_int, _vp, _vp]),
    "bn_forward_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 3 + [_c_int] + [_vp] * 5 + [ctypes.c_float] * 2 + [_vp] * 4
                                + [_c_size, _vp]),
    "bn_backward_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 6 + [_c_int] * 2 + [_vp] * 4 + [_c_size, _vp]),
    "bn_backward_pair_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 6 + [_c_int] * 2 + [_vp] * 4 + [_c_size, _vp]),
    "bn_backward_records_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 6 + [_c_int] * 2 + [_vp] * 4 + [_c_int, _vp]),
    "rows_gemm_bnbwd_hip_launcher": (_c_int, [_c_int] * 4 + [_vp] * 2 + [_c_int] + [_vp] * 6 + [_c_int] + [_vp] * 2),
    "linear_wgrad_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 5 + [_c_size, _vp]),
    "linear_wgrad_multi_hip_launcher": (_c_int, [_c_int] * 4 + [_vp] * 7 + [_c_size, _vp]),
    "skinny_linear_forward_xf_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 6),
    "bn_tiles_floats": (_c_size, [_c_int] * 2),
    "bn_tiles_finalize_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 10 + [ctypes.c_float] * 2 + [_vp]),
    "bn_stats_affine_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 10 + [ctypes.c_float] * 2 + [_vp, _c_size, _vp]),
    "bn_apply_residual_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 9),
    "bn_backward_residual_hip_launcher": (_c_int, [_c_int] * 2 + [_vp] * 7 + [_c_int] + [_vp] * 5 + [_c_size, _vp]),
    "skinny_linear_forward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 4),
    "skinny_linear_backward_hip_launcher": (_c_int, [_c_int] * 3 + [_vp] * 4),
    "linear_wgrad_strided_hip_launcher": (_c_int, [_c_int] * 4 + [_vp, ctypes.c_longlong, ctypes.c_longlong, _vp,
                                                                 ctypes.c_longlong, ctypes.c_longlong, _vp, _vp, _vp,
                                                                 _c_size, _vp]),
}

_ERR = {1: "PTV2_ERR_ARG (invalid argument)", 2: "PTV2_ERR_WORKSPACE (workspace too small)",
        3: "PTV2_ERR_LAUNCH (HIP launch failed)"}
_lib = None
# bumped together with ptv2_abi_version() (ao_amd/csrc/abi.hip) whenever a launcher signature or a struct that ctypes
# mirrors (block.py::_Blk, _BlkGrads) changes: a stale libptv2_hip.so then refuses to load instead of misreading memory
EXPECTED_ABI = 11


def build(verbose=False):
    """Compile every .hip under ao_amd/csrc for gfx950 into ao_amd/lib/libptv2_hip.so (make + hipcc)."""
    cmd = ["make", "-C", CSRC, "-j4", "all"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def register(signatures):
    """Let other csrc units (gva, gridpool, ...) add their entry points to the binding table."""
    _SIGNATURES.update(signatures)
    if _lib is not None:
        _bind(_lib, signatures)


def _bind(lib, signatures):
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "ao_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C ao_amd/csrc` (needs hipcc). There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        _bind(handle, _SIGNATURES)
        have = handle.ptv2_abi_version()
        if have != EXPECTED_ABI:
            raise RuntimeError("ao_amd: %s has ABI version %d, the python side expects %d -- stale build; rebuild with "
                               "`make -C ao_amd/csrc`" % (LIB_PATH, have, EXPECTED_ABI))
        _lib = handle
    return _lib


def check_struct(which, mirror):
    """A ctypes mirror of a C struct must have the size the library was compiled with."""
    have, want = lib().ptv2_struct_bytes(which), ctypes.sizeof(mirror)
    if have != want:
        raise RuntimeError("ao_amd: %s is %d bytes in python, %d in %s (stale build or a drifted mirror)"
                           % (mirror.__name__, want, have, LIB_PATH))


def check(status, what):
    if status != 0:
        raise RuntimeError("ao_amd: %s failed with status %d: %s" % (what, status, _ERR.get(status, "?")))


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ao_amd ops run on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)


def ptr(t):
    return 0 if t is None else t.data_ptr()


_WS = {}


def workspace(nbytes, device):
    """Scratch for one launcher call.  One grow-only buffer per (device, stream): launches on a stream are
    ordered, and no launcher needs its scratch after it returns, so consecutive calls can share it (saves an
    allocator round trip per op; at 288 GB per GPU the retained high-water mark is irrelevant)."""
    nbytes = max(int(nbytes), 256)
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def kernel_timer(enable, only=None, stride=1):
    """Switch the in-library per-kernel HIP-event timer (include/ptv2_hip.h: ptv2_profile_*).  `only` = kernel
    name: bracket that kernel alone (a whole-step measurement is then not perturbed by ~2000 event pairs)."""
    L = lib()
    kid = -1
    if only is not None:
        name = ctypes.create_string_buffer(64)
        us, cnt, byt = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
        for i in range(L.ptv2_profile_kernel_count()):
            L.ptv2_profile_read(i, name, ctypes.byref(us), ctypes.byref(cnt), ctypes.byref(byt))
            if name.value.decode() == only:
                kid = i
        if kid < 0:
            raise ValueError("unknown kernel name %r" % only)
    L.ptv2_profile_select(kid)
    L.ptv2_profile_stride(int(stride))
    L.ptv2_profile_enable(1 if enable else 0)


def graph_stats(reset=False):
    """Counters of the graph-issued model launchers (ao_amd/csrc/graph.hip)."""
    out = (ctypes.c_double * 9)()
    lib().ptv2_graph_stats(out, 1 if reset else 0)
    keys = ("scopes", "updated", "instantiated", "declined", "nodes", "capture_us", "update_us", "launch_us", "wait_us")
    return dict(zip(keys, [float(v) for v in out]))


def kernel_timer_read():
    """{kernel name: dict(launches, total_us, avg_us, bytes_per_launch)} for kernels launched while enabled."""
    L = lib()
    out = {}
    name = ctypes.create_string_buffer(64)
    us, cnt, byt = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
    for kid in range(L.ptv2_profile_kernel_count()):
        if L.ptv2_profile_read(kid, name, ctypes.byref(us), ctypes.byref(cnt), ctypes.byref(byt)) == 0 and cnt.value > 0:
            out[name.value.decode()] = dict(launches=cnt.value, total_us=us.value, avg_us=us.value / cnt.value,
                                            bytes_per_launch=byt.value)
    return out
