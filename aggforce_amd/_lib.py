"""ctypes binding of libaggf.so (the C ABI declared in include/aggf.h).

The library is the product's only compute path: there is no CPU fallback.  If
``libaggf.so`` has not been built (``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C aggforce_amd/csrc``) every compute entry point raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# AGGF_LIB_PATH: tests only -- the adversarial-dispatch-order build of the same sources (tests/test_gpu_order.py)
LIB_PATH = os.environ.get("AGGF_LIB_PATH") or os.path.join(_HERE, "libaggf.so")

F32, F64 = 0, 1
NAN_PROPAGATE, NAN_REPLACE = 0, 1

_lib: Optional[C.CDLL] = None

_vp, _i32, _i64, _u64, _dbl, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/aggf.h one to one
PROTOTYPES = {
    "aggf_version": (C.c_int, []),
    "aggf_last_error": (C.c_char_p, []),
    "aggf_coverage_dump": (_sz, [C.c_char_p, _sz]),
    "aggf_coverage_reset": (C.c_int, []),
    "aggf_device_info": (C.c_int, [C.POINTER(_i32), C.POINTER(_sz), C.POINTER(_sz)]),
    "aggf_gram_workspace_bytes": (_sz, [_i64, _i32, _i32, C.c_int, C.c_int, C.c_int]),
    "aggf_gram": (C.c_int, [_vp, _i64, _i32, C.c_int, C.c_int, _vp, _vp, _i32, _vp, C.c_int, _vp, _sz, _vp]),
    "aggf_gram_from_column": (C.c_int, [_vp, _i64, _i32, C.c_int, C.c_int, _i32, _i32, _vp, C.c_int, _vp, _sz, _vp]),
    "aggf_eq_qp_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "aggf_eq_qp_solve": (C.c_int, [_vp, _i32, _dbl, _vp, _vp, _i32, _vp, _i32, _dbl, _i32, _vp, _vp, _vp, _sz, _vp]),
    "aggf_eq_qp_batched_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "aggf_eq_qp_solve_batched": (C.c_int, [_vp, _i32, _dbl, _vp, _vp, _i32, _vp, _i32, _dbl, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "aggf_eq_qp_solve_batched_shift": (C.c_int, [_vp, _i32, _dbl, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _dbl, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "aggf_eq_qp_pinned_workspace_bytes": (_sz, [_i32, _i32]),
    "aggf_eq_qp_solve_pinned": (C.c_int, [_vp, _i32, _dbl, _vp, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "aggf_expand_map": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "aggf_linearmap_apply_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "aggf_linearmap_apply": (C.c_int, [_vp, _i64, _i32, C.c_int, _vp, _i32, C.c_int, C.c_int, _dbl, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aggf_slice_gather": (C.c_int, [_vp, _i64, _i32, C.c_int, _vp, _i32, C.c_int, _vp, _vp, _vp]),
    "aggf_has_nan": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp]),
    "aggf_not_close": (C.c_int, [_vp, _vp, _i64, C.c_int, _dbl, _dbl, _vp, _vp]),
    "aggf_sumsq_workspace_bytes": (_sz, []),
    "aggf_sumsq": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp, _sz, _vp]),
    "aggf_sym_pack_upper": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp]),
    "aggf_sym_unpack_upper": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp]),
    "aggf_condnormal_augment": (C.c_int, [_vp, _vp, _i64, _i32, C.c_int, _vp, _vp, _vp, _i32, C.c_int, _vp, _vp, _u64, _i64, _dbl, _dbl, _vp, _vp, _vp]),
    "aggf_condnormal_sites": (C.c_int, [_vp, _vp, _u64, _i64, _i64, _i32, C.c_int, _dbl, _dbl, _vp, _vp, C.c_int, _vp]),
    "aggf_residual_over_var": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _i64, _dbl, _vp, _vp, C.c_int, _vp]),
    "aggf_frames_matmul": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _dbl, C.c_int, _vp, _vp]),
    "aggf_augment_concat": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, _i64, _i32, _i32, _dbl, _vp, _vp, _vp]),
    "aggf_gram_pair_workspace_bytes": (_sz, [_i64, _i32, _i32, C.c_int]),
    "aggf_gram_pair": (C.c_int, [_vp, _i32, _vp, _i32, _i64, C.c_int, _vp, C.c_int, _vp, _sz, _vp]),
    "aggf_augmented_gram_workspace_bytes": (_sz, [_i32, _i32]),
    "aggf_augmented_gram": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aggf_sym_group_reduce": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "aggf_group_reduce": (C.c_int, [_vp, _i64, _i32, C.c_int, _vp, _vp, _i32, C.c_int, C.c_int, _vp, _vp]),
    "aggf_gb_channels": (C.c_int, [_vp, _vp, C.c_int, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _dbl, _dbl, _vp, _vp, _vp]),
    "aggf_gb_regmat": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _i64, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _dbl, _dbl, _dbl, _i32, _vp, C.c_int, _vp]),
    "aggf_gb_apply_cols": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _dbl, _dbl, _vp, _vp]),
    "aggf_gb_apply": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _i64, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _dbl, _dbl, _vp, _i32, _vp, _vp]),
    "aggf_trjdot_frames": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _i64, _i32, _i32, _vp, _vp, C.c_int, _vp]),
    "aggf_feat_contract": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _dbl, _i64, _i32, _i32, _i32, _vp, C.c_int, _vp]),
    "aggf_feat_constraint_rows": (C.c_int, [_vp, C.c_int, _i64, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "aggf_gb_constraint_rows": (C.c_int, [_vp, _vp, C.c_int, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "aggf_gb_group_overlap": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "aggf_gb_constraint_gram": (C.c_int, [_vp, _vp, C.c_int, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "aggf_gb_distance_range": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "aggf_gb_regmat_cols": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _dbl, _dbl, _dbl, _i32, _vp, C.c_int, _vp]),
    "aggf_feat_weights": (C.c_int, [_vp, C.c_int, _i64, _i32, _i32, _vp, _i64, _vp, _vp]),
    "aggf_pair_dist_var_workspace_bytes": (_sz, [_i64, _i32]),
    "aggf_pair_dist_var": (C.c_int, [_vp, _i64, _i32, C.c_int, _vp, _vp, _sz, _vp]),
    "aggf_pair_dist_moments": (C.c_int, [_vp, _i64, _i32, C.c_int, _vp, _vp, _vp, _sz, _vp]),
    "aggf_pair_pool_term": (C.c_int, [_vp, _vp, _vp, _dbl, _i64, _vp, _vp]),
    "aggf_gram_quadform_workspace_bytes": (_sz, [_i32, _i32]),
    "aggf_gram_quadform": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
    "aggf_daxpby": (C.c_int, [_i64, _dbl, _vp, _dbl, _vp, _vp, _vp]),
    "aggf_comm_unique_id": (C.c_int, [_vp, _sz]),
    "aggf_comm_init": (C.c_int, [_vp, _sz, _i32, _i32, C.POINTER(_vp)]),
    "aggf_comm_destroy": (C.c_int, [_vp]),
    "aggf_allreduce_sum": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp]),
    "aggf_take_frames": (C.c_int, [_vp, _i64, _i64, C.c_int, _vp, _i64, _vp, _vp]),
    "aggf_concat_sites": (C.c_int, [_vp, _i32, C.c_int, _vp, _i32, C.c_int, _i64, _vp, C.c_int, _vp]),
    "aggf_scale": (C.c_int, [_vp, _i64, C.c_int, _dbl, _vp, _vp]),
    "aggf_synth_normal": (C.c_int, [_vp, _i64, _i32, C.c_int, _u64, _i64, _dbl, _dbl, _dbl, _vp]),
}


class AggfError(RuntimeError):
    """A libaggf call returned an error code."""


def load() -> C.CDLL:
    """Load libaggf.so and set the prototypes (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libaggf.so not found at {LIB_PATH}: the HIP library is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C aggforce_amd/csrc`). "
            "aggforce_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def lib() -> C.CDLL:
    """The loaded library, for compute calls: also requires a visible GPU."""
    l = load()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "aggforce_amd needs an AMD GPU (torch.cuda.is_available() is False); "
            "there is no CPU fallback for the force-map hot path."
        )
    return l


def coverage(names: bool = False, total: bool = False) -> dict:
    """Mangled kernel name -> launches by this process since the last ``aggf_coverage_reset`` (``total=True``: since
    the library was loaded); every kernel the process has ever launched is listed.  ``names=True``: -> (demangled name,
    launches), the demangled name being the string rocprofv3 prints for the kernel."""
    l = load()
    need = l.aggf_coverage_dump(None, 0)
    buf = C.create_string_buffer(need + 1)
    l.aggf_coverage_dump(buf, need + 1)
    out = {}
    for line in buf.value.decode().splitlines():
        mangled, pretty, cnt, tot = line.split("\t")
        n = int(tot) if total else int(cnt)
        out[mangled] = (pretty, n) if names else n
    return out


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().aggf_last_error()
        raise AggfError(f"{what or 'libaggf'} failed (code {rc}): {msg.decode() if msg else ''}")


def dtype_code(dt) -> int:
    if dt in (torch.float32, np.float32) or (not isinstance(dt, torch.dtype) and np.dtype(dt) == np.float32):
        return F32
    if dt in (torch.float64, np.float64) or (not isinstance(dt, torch.dtype) and np.dtype(dt) == np.float64):
        return F64
    raise TypeError(f"unsupported dtype {dt}: libaggf computes in float32 or float64")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device=None) -> int:
    """Raw hipStream_t of torch's current stream on ``device`` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


_ws_cache: dict = {}
_WS_MAX_PER_TAG = 6


def workspace(nbytes: int, device, tag: str = "") -> torch.Tensor:
    """A cached uint8 scratch tensor of at least nbytes (256-byte aligned by the allocator).

    One buffer per (device, tag, current stream): calls issued on different streams never share scratch.
    The cache keeps at most ``_WS_MAX_PER_TAG`` buffers per (device, tag), least recently used first out,
    so work spread over many streams cannot pin HBM for ever."""
    key = (str(device), tag, torch.cuda.current_stream(device).cuda_stream)
    w = _ws_cache.pop(key, None)
    if w is None or w.numel() < nbytes:
        w = None
        w = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    _ws_cache[key] = w  # most recently used last
    same = [k for k in _ws_cache if k[0] == key[0] and k[1] == tag]
    for old in same[: max(0, len(same) - _WS_MAX_PER_TAG)]:
        del _ws_cache[old]
    return w


def free_workspaces() -> None:
    _ws_cache.clear()


def drop_workspace(tag: str, device=None) -> None:
    """Forget the cached scratch buffers of one tag (on one device, or everywhere)."""
    for key in [k for k in _ws_cache if k[1] == tag and (device is None or k[0] == str(device))]:
        del _ws_cache[key]
