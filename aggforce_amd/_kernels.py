"""Thin Python wrappers over the libaggf C ABI, operating on torch (ROCm) tensors.

torch is only the array container here (device memory, current stream); all
arithmetic happens in the hand-written HIP kernels behind ``include/aggf.h``.
"""
from __future__ import annotations

import contextlib
import weakref
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, drop_workspace, dtype_code, lib, ptr, stream_ptr, workspace  # noqa: F401

_TORCH_OF = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}
_NP_OF = {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64)}

# ------------------------------------------------------------------ per-stage HIP-event timers

_timers: Optional[dict] = None


def start_timers() -> None:
    """Bracket every libaggf stage with HIP events on the launching stream (bench.py)."""
    global _timers
    _timers = {}


def stop_timers() -> dict:
    """{stage: {"ms": total, "calls": n}}; synchronises the device."""
    global _timers
    out = {}
    if _timers is not None:
        torch.cuda.synchronize()
        for name, pairs in _timers.items():
            out[name] = {"ms": float(sum(a.elapsed_time(b) for a, b in pairs)), "calls": len(pairs)}
    _timers = None
    return out


@contextlib.contextmanager
def _timed(name: str):
    if _timers is None:
        yield
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    try:
        yield
    finally:
        b.record()
        _timers.setdefault(name, []).append((a, b))


# ------------------------------------------------------------------ containers

_cache_stack: list = []


@contextlib.contextmanager
def upload_cache():
    """Within this context a NumPy array is copied to the GPU at most once (keyed by identity)."""
    _cache_stack.append({})
    try:
        yield
    finally:
        _cache_stack.pop()


def is_torch(x) -> bool:
    return isinstance(x, torch.Tensor)


def default_device() -> torch.device:
    lib()
    return torch.device("cuda", torch.cuda.current_device())


def as_device(x, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """NumPy array / CPU tensor / GPU tensor -> contiguous GPU tensor (float32 or float64)."""
    if is_torch(x):
        t = x
        if t.dtype not in (torch.float32, torch.float64):
            t = t.to(torch.float64)
        if not t.is_cuda:
            t = t.to(default_device())
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous()
    arr = np.asarray(x)
    key = None
    if _cache_stack and isinstance(x, np.ndarray):
        key = (id(x), str(dtype))
        hit = _cache_stack[-1].get(key)
        if hit is not None and hit[0]() is x:
            return hit[1]
    if arr.dtype not in (np.float32, np.float64):
        arr = arr.astype(np.float64)
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(default_device())
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if key is not None:
        _cache_stack[-1][key] = (weakref.ref(x), t)
    return t


def like_input(t: torch.Tensor, template):
    """Return t as the same kind of container as template (NumPy array or torch tensor)."""
    if is_torch(template):
        return t if template.is_cuda else t.cpu()
    return t.cpu().numpy()


def np_dtype_of(x) -> np.dtype:
    if is_torch(x):
        return _NP_OF.get(x.dtype, np.dtype(np.float64))
    return np.asarray(x).dtype if not isinstance(x, np.ndarray) else x.dtype


def torch_dtype(npdt) -> torch.dtype:
    return _TORCH_OF[np.dtype(npdt)]


# ------------------------------------------------------------------ K1 Gram


def gram(
    forces: torch.Tensor,
    grp_ptr: Optional[torch.Tensor],
    grp_atoms: Optional[torch.Tensor],
    n_red: int,
    compute_dtype: torch.dtype,
    out: Optional[torch.Tensor] = None,
    accumulate: bool = False,
    ws_limit_bytes: Optional[int] = None,
    first_col: int = 0,
) -> torch.Tensor:
    """G (n_red, n_red) float64 from forces (T, N, 3); see aggf_gram in include/aggf.h.  ``first_col`` > 0
    (a multiple of 128, no constraint groups): the leading first_col x first_col block of ``out`` is the
    caller's and is not computed (aggf_gram_from_column)."""
    l = lib()
    T, N, D = forces.shape
    if D != 3:
        raise ValueError("forces must have shape (n_frames, n_sites, 3)")
    if T == 0:
        raise ValueError("empty trajectory")
    dev = forces.device
    if out is None:
        out = torch.empty((n_red, n_red), dtype=torch.float64, device=dev)
        accumulate = False
    ind, cd = dtype_code(forces.dtype), dtype_code(compute_dtype)
    need = l.aggf_gram_workspace_bytes(T, N, n_red, ind, cd, 1 if grp_ptr is not None else 0)
    if ws_limit_bytes is not None:
        need = min(need, int(ws_limit_bytes))
    ws = workspace(need, dev, "gram")
    with _timed("gram"):
        if first_col:
            if grp_ptr is not None:
                raise ValueError("first_col needs a regression matrix without constraint groups")
            check(
                l.aggf_gram_from_column(ptr(forces), T, N, ind, cd, n_red, int(first_col), ptr(out),
                                        1 if accumulate else 0, ptr(ws), need, stream_ptr()),
                "aggf_gram_from_column",
            )
        else:
            check(
                l.aggf_gram(ptr(forces), T, N, ind, cd, ptr(grp_ptr), ptr(grp_atoms), n_red, ptr(out),
                            1 if accumulate else 0, ptr(ws), need, stream_ptr()),
                "aggf_gram",
            )
    return out


def eq_qp_solve(
    G: torch.Tensor,
    l2: float,
    l2_diag: Optional[torch.Tensor],
    A: torch.Tensor,
    B: Optional[torch.Tensor] = None,
    schur_reg: float = 0.0,
    n_refine: int = 1,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """X (nrhs, n) and stats (4,) -- both on the device; see aggf_eq_qp_solve."""
    l = lib()
    n = G.shape[0]
    m = A.shape[0]
    nrhs = m if B is None else B.shape[1]
    dev = G.device
    X = torch.empty((nrhs, n), dtype=torch.float64, device=dev)
    stats = torch.empty(4, dtype=torch.float64, device=dev)
    need = l.aggf_eq_qp_workspace_bytes(n, m, nrhs)
    ws = workspace(need, dev, "solve")
    with _timed("solve"):
        check(
            l.aggf_eq_qp_solve(ptr(G), n, float(l2), ptr(l2_diag), ptr(A), m, ptr(B), nrhs, float(schur_reg),
                               int(n_refine), ptr(X), ptr(stats), ptr(ws), need, stream_ptr()),
            "aggf_eq_qp_solve",
        )
    return X, stats


def eq_qp_solve_pinned(G: torch.Tensor, l2: float, l2_diag: Optional[torch.Tensor], pin_idx: torch.Tensor):
    """X (m, n) and stats (4,) for one-hot constraint rows: row i of the constraint matrix is the unit vector at
    pin_idx[i] (int32 device array, distinct entries), right-hand sides = identity; see aggf_eq_qp_solve_pinned."""
    l = lib()
    n, m = G.shape[0], pin_idx.numel()
    X = torch.empty((m, n), dtype=torch.float64, device=G.device)
    stats = torch.empty(4, dtype=torch.float64, device=G.device)
    need = l.aggf_eq_qp_pinned_workspace_bytes(n, m)
    ws = workspace(need, G.device, "solve")
    with _timed("solve"):
        check(l.aggf_eq_qp_solve_pinned(ptr(G), n, float(l2), ptr(l2_diag), ptr(pin_idx), m, ptr(X), ptr(stats), ptr(ws),
                                        need, stream_ptr()), "aggf_eq_qp_solve_pinned")
    return X, stats


def eq_qp_batched_bytes(n: int, m: int, nrhs: int, n_problems: int) -> int:
    return int(lib().aggf_eq_qp_batched_workspace_bytes(n, m, nrhs, n_problems))


def eq_qp_solve_batched(
    G: torch.Tensor,
    l2: float,
    l2_diag: Optional[torch.Tensor],
    A: torch.Tensor,
    B: Optional[torch.Tensor] = None,
    schur_reg: float = 0.0,
    n_refine: int = 1,
    AtA: Optional[torch.Tensor] = None,
    perm: Optional[torch.Tensor] = None,
    a_first_col: int = 0,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Independent problems side by side: G (p, n, n), A (p, m, n), B (p, m, nrhs) or None ->
    X (p, nrhs, n), stats (p, 4); see aggf_eq_qp_solve_batched.  ``AtA`` (p, n, n): the caller's A'A (lower
    triangle read), aggf_eq_qp_solve_batched_shift; with it ``perm`` (p, n) int32 = the order in which the
    factorisation takes the variables and ``a_first_col`` = the number of leading variables (in that order) no
    constraint row touches in any problem."""
    l = lib()
    if G.dim() != 3 or A.dim() != 3 or G.shape[0] != A.shape[0] or G.shape[1] != G.shape[2] or A.shape[2] != G.shape[1]:
        raise ValueError(f"shape mismatch: G {tuple(G.shape)}, A {tuple(A.shape)}")
    if G.dtype != torch.float64 or A.dtype != torch.float64 or not G.is_contiguous() or not A.is_contiguous():
        raise ValueError("G and A must be contiguous float64")
    npb, n, _ = G.shape
    m = A.shape[1]
    if B is not None and (B.dim() != 3 or B.shape[0] != npb or B.shape[1] != m or B.dtype != torch.float64
                          or not B.is_contiguous()):
        raise ValueError(f"shape mismatch: B {tuple(B.shape)}")
    nrhs = m if B is None else B.shape[2]
    dev = G.device
    X = torch.empty((npb, nrhs, n), dtype=torch.float64, device=dev)
    stats = torch.empty((npb, 4), dtype=torch.float64, device=dev)
    need = l.aggf_eq_qp_batched_workspace_bytes(n, m, nrhs, npb)
    ws = workspace(need, dev, "solve")
    if AtA is not None:
        if AtA.shape != G.shape or AtA.dtype != torch.float64 or not AtA.is_contiguous():
            raise ValueError(f"AtA must be contiguous float64 of G's shape, got {tuple(AtA.shape)}")
        if perm is not None and (perm.shape != (npb, n) or perm.dtype != torch.int32 or not perm.is_contiguous()):
            raise ValueError(f"perm must be contiguous int32 of shape {(npb, n)}, got {tuple(perm.shape)}")
        with _timed("solve"):
            check(
                l.aggf_eq_qp_solve_batched_shift(ptr(G), n, float(l2), ptr(l2_diag), ptr(A), ptr(AtA), ptr(perm),
                                                 int(a_first_col) if perm is not None else 0, m, ptr(B), nrhs,
                                                 float(schur_reg), int(n_refine), npb, ptr(X), ptr(stats), ptr(ws), need,
                                                 stream_ptr()),
                "aggf_eq_qp_solve_batched_shift",
            )
        return X, stats
    if perm is not None:
        raise ValueError("perm needs AtA (aggf_eq_qp_solve_batched_shift)")
    with _timed("solve"):
        check(
            l.aggf_eq_qp_solve_batched(ptr(G), n, float(l2), ptr(l2_diag), ptr(A), m, ptr(B), nrhs, float(schur_reg),
                                       int(n_refine), npb, ptr(X), ptr(stats), ptr(ws), need, stream_ptr()),
            "aggf_eq_qp_solve_batched",
        )
    return X, stats


def device_memory(device=None) -> Tuple[int, int]:
    """(free, total) bytes of HBM on the current device (aggf_device_info)."""
    import ctypes as C

    cu, free, total = C.c_int32(0), C.c_size_t(0), C.c_size_t(0)
    check(lib().aggf_device_info(C.byref(cu), C.byref(free), C.byref(total)), "aggf_device_info")
    return int(free.value), int(total.value)


def expand_map(X: torch.Tensor, group_of_atom: torch.Tensor, N: int) -> torch.Tensor:
    l = lib()
    n_rows, n_red = X.shape
    W = torch.empty((n_rows, N), dtype=torch.float64, device=X.device)
    check(l.aggf_expand_map(ptr(X), n_rows, n_red, ptr(group_of_atom), N, ptr(W), stream_ptr()), "aggf_expand_map")
    return W


# ------------------------------------------------------------------ K3 apply


def linearmap_apply(
    points: torch.Tensor,
    matrix: torch.Tensor,
    nan_fill: Optional[float] = None,
    want_sumsq: bool = False,
    nan_probe: Optional[torch.Tensor] = None,
):
    """out (T, n_cg, 3) in matrix.dtype; optional device scalar sum of squares.

    ``nan_probe``: zeroed int32 device scalar that is set to 1 if ``points`` holds a NaN."""
    l = lib()
    T, N, D = points.shape
    n_cg, N2 = matrix.shape
    if D != 3 or N2 != N:
        raise ValueError(f"shape mismatch: points {tuple(points.shape)}, matrix {tuple(matrix.shape)}")
    dev = points.device
    out = torch.empty((T, n_cg, 3), dtype=matrix.dtype, device=dev)
    if T == 0:
        return (out, torch.zeros(1, dtype=torch.float64, device=dev)) if want_sumsq else out
    sumsq = torch.empty(1, dtype=torch.float64, device=dev) if want_sumsq else None
    need = l.aggf_linearmap_apply_workspace_bytes(T, N, n_cg) if want_sumsq else 0
    ws = workspace(need, dev, "apply") if want_sumsq else None
    with _timed("apply"):
        check(
            l.aggf_linearmap_apply(ptr(points), T, N, dtype_code(points.dtype), ptr(matrix), n_cg,
                                   dtype_code(matrix.dtype),
                                   _lib.NAN_REPLACE if nan_fill is not None else _lib.NAN_PROPAGATE,
                                   0.0 if nan_fill is None else float(nan_fill), ptr(out), ptr(sumsq), ptr(nan_probe),
                                   ptr(ws), need, stream_ptr()),
            "aggf_linearmap_apply",
        )
    return (out, sumsq) if want_sumsq else out


def slice_gather(points: torch.Tensor, idx: torch.Tensor, out_dtype: torch.dtype,
                 nan_probe: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[t, c, :] = points[t, idx[c], :] (aggf_slice_gather).  ``nan_probe``: zeroed int32 device scalar (from
    :func:`take_flag`) that is set to 1 if a gathered value is NaN -- the scan rides along with the gather."""
    l = lib()
    T, N, D = points.shape
    n_cg = idx.numel()
    out = torch.empty((T, n_cg, 3), dtype=out_dtype, device=points.device)
    if T == 0:
        return out
    with _timed("gather"):
        check(
            l.aggf_slice_gather(ptr(points), T, N, dtype_code(points.dtype), ptr(idx), n_cg, dtype_code(out_dtype),
                                ptr(out), ptr(nan_probe), stream_ptr()),
            "aggf_slice_gather",
        )
    return out


_flag_pool: dict = {}


def take_flag(device) -> torch.Tensor:
    """A ZEROED int32 device scalar (shape (1,)) from a per-device pool; hand it back with :func:`read_flag`.

    Why a pool: ``torch.zeros(1)`` is a fill kernel on the current stream, and a one-workgroup kernel queued while a
    grid-stride kernel of another stream (the coordinate gather underneath the force map's apply) owns every wave
    slot waits until one of that kernel's workgroups retires -- 1.4 ms + 0.55 ms per step at BASELINE's
    configuration (rocprofv3 timeline, round 3).  Pool flags are zeroed in blocks of 64 and re-zeroed only after a
    kernel has actually set them, so the steady state issues no fill at all."""
    pool = _flag_pool.setdefault(str(device), [])
    if not pool:
        block = torch.zeros(64, dtype=torch.int32, device=device)
        # The fill runs on whatever stream is current (in a fresh process: the side stream of the coordinate gather,
        # behind a multi-millisecond kernel) while the flags are then used on any stream: wait for it here, once per
        # 64 flags, so that no later kernel's mark can be wiped by -- or read before -- the fill.
        torch.cuda.current_stream(block.device).synchronize()
        pool.extend(block[i:i + 1] for i in range(64))
    return pool.pop()


def read_flag(flag: torch.Tensor) -> bool:
    """Host value of a flag from :func:`take_flag` (synchronises); the flag goes back to the pool, zeroed.  A flag is
    handed back exactly once: a second hand-back would let two later kernels share it."""
    pool = _flag_pool.setdefault(str(flag.device), [])
    if any(f.data_ptr() == flag.data_ptr() for f in pool):
        raise RuntimeError("read_flag: this flag is already back in the pool")
    v = bool(flag.item())
    if v:
        flag.zero_()
        torch.cuda.current_stream(flag.device).synchronize()  # the next user may sit on another stream
    pool.append(flag)
    return v


def nan_flag(x: torch.Tensor) -> torch.Tensor:
    """Device int32 flag (shape (1,), from :func:`take_flag`: read it with :func:`read_flag`): non-zero iff x holds a
    NaN.  No host synchronisation."""
    flag = take_flag(x.device)
    if x.numel():
        check(lib().aggf_has_nan(ptr(x), x.numel(), dtype_code(x.dtype), ptr(flag), stream_ptr()), "aggf_has_nan")
    return flag


_side_streams: dict = {}


def side_stream(device) -> "torch.cuda.Stream":
    """One auxiliary stream per device for work that may overlap the main stream's kernels."""
    key = str(device)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[key] = st
    return st


def side_streams(device, n: int) -> list:
    """n auxiliary streams per device, created once (workspaces are cached per stream: fresh streams on
    every call would pin a new set of scratch buffers each time)."""
    key = (str(device), "pool")
    pool = _side_streams.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def has_nan(x: torch.Tensor) -> bool:
    l = lib()
    if x.numel() == 0:
        return False
    flag = take_flag(x.device)
    check(l.aggf_has_nan(ptr(x), x.numel(), dtype_code(x.dtype), ptr(flag), stream_ptr()), "aggf_has_nan")
    return read_flag(flag)


def allclose(a: torch.Tensor, b: torch.Tensor, rtol: float = 1e-5, atol: float = 1e-8) -> bool:
    l = lib()
    if a.numel() == 0:
        return True
    flag = take_flag(a.device)
    check(
        l.aggf_not_close(ptr(a), ptr(b), a.numel(), dtype_code(a.dtype), float(rtol), float(atol), ptr(flag),
                         stream_ptr()),
        "aggf_not_close",
    )
    return not read_flag(flag)


def sumsq(x: torch.Tensor) -> torch.Tensor:
    """Device scalar (shape (1,), float64): sum of squares of x, fixed summation order."""
    l = lib()
    out = torch.zeros(1, dtype=torch.float64, device=x.device)
    if x.numel() == 0:
        return out
    need = l.aggf_sumsq_workspace_bytes()
    ws = workspace(need, x.device, "sumsq")
    check(l.aggf_sumsq(ptr(x), x.numel(), dtype_code(x.dtype), ptr(out), ptr(ws), need, stream_ptr()), "aggf_sumsq")
    return out


def sym_pack_upper(G: torch.Tensor) -> torch.Tensor:
    """Packed upper triangles (batch, n (n + 1) / 2) of symmetric float64 matrices G (..., n, n)."""
    l = lib()
    n = G.shape[-1]
    assert G.dtype == torch.float64 and G.shape[-2] == n and G.is_contiguous()
    batch = G.numel() // (n * n)
    out = torch.empty((batch, n * (n + 1) // 2), dtype=torch.float64, device=G.device)
    check(l.aggf_sym_pack_upper(ptr(G), n, batch, ptr(out), stream_ptr(G.device)), "aggf_sym_pack_upper")
    return out


def sym_unpack_upper(packed: torch.Tensor, G: torch.Tensor) -> torch.Tensor:
    """Both triangles of G (..., n, n) from packed upper triangles; in place, returns G."""
    l = lib()
    n = G.shape[-1]
    assert G.dtype == torch.float64 and packed.dtype == torch.float64 and G.is_contiguous() and packed.is_contiguous()
    batch = G.numel() // (n * n)
    assert packed.numel() == batch * (n * (n + 1) // 2)
    check(l.aggf_sym_unpack_upper(ptr(packed), n, batch, ptr(G), stream_ptr(G.device)), "aggf_sym_unpack_upper")
    return G


def gram_quadform(G: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    """q[i] = x_i' G x_i for the rows of X (m, n); float64 on the device."""
    l = lib()
    n = G.shape[0]
    m = X.shape[0]
    assert G.dtype == torch.float64 and X.dtype == torch.float64 and X.shape[1] == n
    q = torch.empty(m, dtype=torch.float64, device=G.device)
    need = l.aggf_gram_quadform_workspace_bytes(n, m)
    ws = workspace(need, G.device, "quadform")
    check(l.aggf_gram_quadform(ptr(G.contiguous()), n, ptr(X.contiguous()), m, ptr(q), ptr(ws), need, stream_ptr()),
          "aggf_gram_quadform")
    return q


def axpby(a: float, x: torch.Tensor, b: float, y: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a*x + b*y (float64, same shape)."""
    assert x.dtype == torch.float64 and y.dtype == torch.float64 and x.shape == y.shape
    x = x.contiguous()
    y = y.contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(lib().aggf_daxpby(x.numel(), float(a), ptr(x), float(b), ptr(y), ptr(out), stream_ptr()), "aggf_daxpby")
    return out


# ------------------------------------------------------------------ K5 augment


def premap_columns(matrix_host: np.ndarray, dtype: torch.dtype, device):
    """Columns of the premap M (n_cg, N) in compressed form for aggf_condnormal_augment."""
    mt = np.ascontiguousarray(matrix_host.T)  # (N, n_cg)
    rows, cols = np.nonzero(mt)
    ptr_h = np.zeros(mt.shape[0] + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=mt.shape[0]), out=ptr_h[1:])
    return (torch.from_numpy(ptr_h).to(device), torch.from_numpy(cols.astype(np.int32)).to(device),
            torch.from_numpy(mt[rows, cols]).to(device=device, dtype=dtype))


def condnormal_augment(
    coords: torch.Tensor,
    forces: torch.Tensor,
    columns,
    n_cg: int,
    mean: torch.Tensor,
    var: float,
    kbt: float,
    noise: Optional[torch.Tensor],
    seed: int,
    frame_offset: int,
):
    l = lib()
    T, N, _ = coords.shape
    mt_ptr, mt_idx, mt_val = columns
    out_dtype = torch.promote_types(coords.dtype, mt_val.dtype)
    oc = torch.empty((T, N + n_cg, 3), dtype=out_dtype, device=coords.device)
    of = torch.empty((T, N + n_cg, 3), dtype=out_dtype, device=coords.device)
    if T == 0:
        return oc, of
    with _timed("augment"):
        check(
            l.aggf_condnormal_augment(ptr(coords), ptr(forces), T, N, dtype_code(coords.dtype), ptr(mt_ptr),
                                      ptr(mt_idx), ptr(mt_val), n_cg, dtype_code(mt_val.dtype), ptr(mean), ptr(noise),
                                      int(seed) & (2**64 - 1), int(frame_offset), float(var), float(kbt), ptr(oc),
                                      ptr(of), stream_ptr()),
            "aggf_condnormal_augment",
        )
    return oc, of


def condnormal_sites(mean: torch.Tensor, var: float, kbt: float, noise: Optional[torch.Tensor], seed: int,
                     frame_offset: int, out_dtype: torch.dtype):
    """(y, Fa): generated-site coordinates and forces (T, n_cg, 3), the non-copy part of the extended trajectory;
    same arithmetic and Philox stream as :func:`condnormal_augment` (aggf_condnormal_sites)."""
    T, n_cg, _ = mean.shape
    y = torch.empty((T, n_cg, 3), dtype=out_dtype, device=mean.device)
    fa = torch.empty((T, n_cg, 3), dtype=out_dtype, device=mean.device)
    if T == 0:
        return y, fa
    with _timed("augment"):
        check(lib().aggf_condnormal_sites(ptr(mean), ptr(noise), int(seed) & (2**64 - 1), int(frame_offset), T, n_cg,
                                          dtype_code(mean.dtype), float(var), float(kbt), ptr(y), ptr(fa),
                                          dtype_code(out_dtype), stream_ptr()), "aggf_condnormal_sites")
    return y, fa


def residual_over_var(gen: torch.Tensor, mean: torch.Tensor, var: float, out_dtype: torch.dtype, want_pos: bool = True,
                      want_neg: bool = True):
    """(r, -r) with r = (gen - mean) / var, elementwise, in ``out_dtype`` (aggf_residual_over_var): the log-gradients
    of a scalar-covariance conditional normal.  An output that is not wanted is None."""
    assert gen.shape == mean.shape and gen.is_contiguous() and mean.is_contiguous()
    pos = torch.empty(gen.shape, dtype=out_dtype, device=gen.device) if want_pos else None
    neg = torch.empty(gen.shape, dtype=out_dtype, device=gen.device) if want_neg else None
    if gen.numel():
        with _timed("augment"):
            check(lib().aggf_residual_over_var(ptr(gen), dtype_code(gen.dtype), ptr(mean), dtype_code(mean.dtype),
                                               gen.numel(), float(var), ptr(pos), ptr(neg), dtype_code(out_dtype),
                                               stream_ptr()), "aggf_residual_over_var")
    return pos, neg


def frames_matmul(x: torch.Tensor, b: torch.Tensor, sub: Optional[torch.Tensor] = None,
                  add: Optional[torch.Tensor] = None, alpha: float = 1.0) -> torch.Tensor:
    """``add + alpha (x - sub) @ b.T`` on flattened frames: x, sub (T, K), b (J, K), add (T, J), one dtype
    (aggf_frames_matmul)."""
    T, Kd = x.shape
    J = b.shape[0]
    assert b.shape[1] == Kd and b.dtype == x.dtype and x.is_contiguous() and b.is_contiguous()
    assert sub is None or (sub.shape == x.shape and sub.dtype == x.dtype and sub.is_contiguous())
    assert add is None or (tuple(add.shape) == (T, J) and add.dtype == x.dtype and add.is_contiguous())
    out = torch.empty((T, J), dtype=x.dtype, device=x.device)
    if T:
        with _timed("augment"):
            check(lib().aggf_frames_matmul(ptr(x), ptr(sub), T, Kd, ptr(b), J, ptr(add), float(alpha),
                                           dtype_code(x.dtype), ptr(out), stream_ptr()), "aggf_frames_matmul")
    return out


def augment_concat(coords: torch.Tensor, forces: torch.Tensor, gen: torch.Tensor, corr: torch.Tensor,
                   lgrad: torch.Tensor, kbt: float):
    """([coords ; gen], [forces + kbt corr ; kbt lgrad]) in the promoted dtype (aggf_augment_concat): the
    concatenation step of the general Augmenter protocol (trajectory/core.py:384-390)."""
    T, N, _ = coords.shape
    n_aug = gen.shape[1]
    assert forces.shape == coords.shape and forces.dtype == coords.dtype
    assert gen.dtype == corr.dtype == lgrad.dtype and corr.shape == coords.shape and lgrad.shape == gen.shape
    out_dtype = torch.promote_types(coords.dtype, gen.dtype)
    oc = torch.empty((T, N + n_aug, 3), dtype=out_dtype, device=coords.device)
    of = torch.empty((T, N + n_aug, 3), dtype=out_dtype, device=coords.device)
    if T:
        with _timed("augment"):
            check(lib().aggf_augment_concat(ptr(coords), ptr(forces), dtype_code(coords.dtype), ptr(gen), ptr(corr),
                                            ptr(lgrad), dtype_code(gen.dtype), T, N, n_aug, float(kbt), ptr(oc),
                                            ptr(of), stream_ptr()), "aggf_augment_concat")
    return oc, of


def gram_pair_ok(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Can aggf_gram_pair read [a | b] in place?  (same dtype, 128-multiples of sites, 16-byte aligned)"""
    return (a.dtype == b.dtype and a.dtype in (torch.float32, torch.float64) and a.shape[0] == b.shape[0] and a.shape[0] > 0
            and a.shape[1] % 128 == 0 and b.shape[1] % 128 == 0 and a.is_contiguous() and b.is_contiguous()
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)


def gram_pair(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Gram matrix ((Na + Nb)^2 float64) of the column-concatenation [a | b] of two (T, ., 3) arrays, read where
    they lie (aggf_gram_pair); products in the arrays' dtype."""
    l = lib()
    T, N, _ = a.shape
    N2 = b.shape[1]
    out = torch.empty((N + N2, N + N2), dtype=torch.float64, device=a.device)
    need = l.aggf_gram_pair_workspace_bytes(T, N, N2, dtype_code(a.dtype))
    ws = workspace(need, a.device, "gram")
    with _timed("gram"):
        check(l.aggf_gram_pair(ptr(a), N, ptr(b), N2, T, dtype_code(a.dtype), ptr(out), 0, ptr(ws), need, stream_ptr()),
              "aggf_gram_pair")
    return out


def augmented_gram(Gx: torch.Tensor, N: int, columns) -> torch.Tensor:
    """Tm' Gx Tm, Tm = [[I, 0], [-C, I]]: Gram matrix of [F - Fa C | Fa] from that of [F | Fa] (aggf_augmented_gram);
    ``columns`` = premap_columns(C, float64)."""
    l = lib()
    n = Gx.shape[0]
    n2 = n - N
    cp, ci, cv = columns
    assert Gx.dtype == torch.float64 and cv.dtype == torch.float64 and Gx.is_contiguous()
    out = torch.empty_like(Gx)
    need = l.aggf_augmented_gram_workspace_bytes(N, n2)
    ws = workspace(need, Gx.device, "auggram")
    check(l.aggf_augmented_gram(ptr(Gx), N, n2, ptr(cp), ptr(ci), ptr(cv), ptr(out), ptr(ws), need, stream_ptr()),
          "aggf_augmented_gram")
    return out


def sym_group_reduce(G: torch.Tensor, grp_ptr: torch.Tensor, grp_atoms: torch.Tensor, n_red: int) -> torch.Tensor:
    """C' G C for the 0/1 constraint matrix given as CSR groups (aggf_sym_group_reduce)."""
    out = torch.empty((n_red, n_red), dtype=torch.float64, device=G.device)
    check(lib().aggf_sym_group_reduce(ptr(G.contiguous()), G.shape[0], ptr(grp_ptr), ptr(grp_atoms), n_red, ptr(out),
                                      stream_ptr()), "aggf_sym_group_reduce")
    return out


# ------------------------------------------------------------------ synthetic data


def take_frames(x: torch.Tensor, idx) -> torch.Tensor:
    """x[idx] along the frame axis as one gather kernel (aggf_take_frames); ``idx``: integer array-like or tensor."""
    x = x.contiguous()
    n_src = x.shape[0]
    if not isinstance(idx, torch.Tensor) or not idx.is_cuda:
        # host index array (the fold bookkeeping of the cross-validation): validated and wrapped on the host, where it
        # lies -- no device min / max, no synchronisation inside the fold loops
        host = np.asarray(idx.numpy() if isinstance(idx, torch.Tensor) else idx, dtype=np.int64).reshape(-1)
        if host.size and (host.min() < -n_src or host.max() >= n_src):
            raise IndexError(f"frame index out of range for {n_src} frames")
        if host.size and host.min() < 0:
            host = np.where(host < 0, host + n_src, host)
        idx = torch.from_numpy(np.ascontiguousarray(host)).to(x.device)
    else:
        idx = idx.to(dtype=torch.int64).contiguous()
        lo, hi = (int(v) for v in torch.stack([idx.min(), idx.max()]).tolist()) if idx.numel() else (0, 0)  # one sync
        if idx.numel() and (lo < -n_src or hi >= n_src):
            raise IndexError(f"frame index out of range for {n_src} frames")
        if idx.numel() and lo < 0:
            idx = torch.where(idx < 0, idx + n_src, idx)
    n = idx.numel()
    out = torch.empty((n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if n and n_src:
        row = int(np.prod(x.shape[1:])) if x.dim() > 1 else 1
        check(lib().aggf_take_frames(ptr(x), n_src, row, dtype_code(x.dtype), ptr(idx), n, ptr(out), stream_ptr()),
              "aggf_take_frames")
    return out


def concat_sites(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[a ; b] along the site axis of two (T, ., 3) arrays, in the promoted dtype (aggf_concat_sites)."""
    assert a.dim() == 3 and b.dim() == 3 and a.shape[0] == b.shape[0] and a.shape[2] == 3 and b.shape[2] == 3
    a, b = a.contiguous(), b.contiguous()
    dt = torch.promote_types(a.dtype, b.dtype)
    T = a.shape[0]
    out = torch.empty((T, a.shape[1] + b.shape[1], 3), dtype=dt, device=a.device)
    if T and a.shape[1] and b.shape[1]:
        check(lib().aggf_concat_sites(ptr(a), a.shape[1], dtype_code(a.dtype), ptr(b), b.shape[1], dtype_code(b.dtype), T,
                                      ptr(out), dtype_code(dt), stream_ptr()), "aggf_concat_sites")
    elif T:
        out.copy_(a if a.shape[1] else b)
    return out


def scale(x: torch.Tensor, alpha: float) -> torch.Tensor:
    """alpha * x as a product (aggf_scale): NaN / inf behave as in NumPy's ``alpha * x``."""
    x = x.contiguous()
    out = torch.empty_like(x)
    if x.numel():
        check(lib().aggf_scale(ptr(x), x.numel(), dtype_code(x.dtype), float(alpha), ptr(out), stream_ptr()), "aggf_scale")
    return out


def scaled(x, alpha):
    """``alpha * x`` for a trajectory array: device tensors through :func:`scale`, host arrays as they are (NumPy)."""
    if is_torch(x) and x.is_cuda and x.dtype in (torch.float32, torch.float64):
        return scale(x, float(alpha))
    return alpha * x


def synth_normal(T: int, N: int, dtype: torch.dtype, seed: int, frame_offset: int = 0, mean: float = 0.0,
                 sigma: float = 1.0, lattice: float = 0.0, device=None) -> torch.Tensor:
    l = lib()
    dev = device or default_device()
    out = torch.empty((T, N, 3), dtype=dtype, device=dev)
    check(
        l.aggf_synth_normal(ptr(out), T, N, dtype_code(dtype), int(seed), int(frame_offset), float(mean),
                            float(sigma), float(lattice), stream_ptr()),
        "aggf_synth_normal",
    )
    return out


# ------------------------------------------------------------------ K4 gb_feat


def group_reduce(x: torch.Tensor, grp_ptr: torch.Tensor, grp_atoms: torch.Tensor, n_groups: int, mean: bool,
                 out_dtype: torch.dtype) -> torch.Tensor:
    """(T, n_groups, 3): per-group sums (or means) of x (T, N, 3); see aggf_group_reduce."""
    l = lib()
    T, N, _ = x.shape
    out = torch.empty((T, n_groups, 3), dtype=out_dtype, device=x.device)
    if T == 0:
        return out
    check(
        l.aggf_group_reduce(ptr(x), T, N, dtype_code(x.dtype), ptr(grp_ptr), ptr(grp_atoms), n_groups,
                            1 if mean else 0, dtype_code(out_dtype), ptr(out), stream_ptr()),
        "aggf_group_reduce",
    )
    return out


def _gb_dtype(Pg, cg, centers) -> int:
    """Feature arithmetic type of the K4 calls: Pg, cg and centers must share it (float32 or float64)."""
    if not (Pg.dtype == cg.dtype == centers.dtype):
        raise TypeError(f"gb_feat operands differ in dtype: {Pg.dtype}, {cg.dtype}, {centers.dtype}")
    return dtype_code(Pg.dtype)


def gb_channels(Pg, cg, site: int, sizes, n_ch: int, centers, width: float, clip: float):
    l = lib()
    T, G, _ = Pg.shape
    nb = centers.numel()
    gd = _gb_dtype(Pg, cg, centers)
    gauss = torch.empty((T, n_ch, nb), dtype=Pg.dtype, device=Pg.device)
    grad = torch.empty((T, n_ch, nb, 3), dtype=Pg.dtype, device=Pg.device)
    if T == 0 or n_ch == 0:
        return gauss, grad
    check(
        l.aggf_gb_channels(ptr(Pg), ptr(cg), gd, T, G, cg.shape[1], site, ptr(sizes), n_ch, ptr(centers), nb,
                           float(width), float(clip), ptr(gauss), ptr(grad), stream_ptr()),
        "aggf_gb_channels",
    )
    return gauss, grad


def gb_regmat(Fg, Pg, cg, site: int, sizes, n_id: int, n_ch: int, centers, width: float, clip: float, kbt: float,
              out: torch.Tensor) -> torch.Tensor:
    """Fill out (T, ld_feat, 3) with the regression matrix of one cg site; see aggf_gb_regmat."""
    l = lib()
    T, G, _ = Fg.shape
    with _timed("gb_regmat"):
        check(
            l.aggf_gb_regmat(ptr(Fg), dtype_code(Fg.dtype), ptr(Pg), ptr(cg), _gb_dtype(Pg, cg, centers), T, G,
                             cg.shape[1], site, ptr(sizes),
                             n_id, n_ch, ptr(centers), centers.numel(), float(width), float(clip), float(kbt),
                             out.shape[1], ptr(out), dtype_code(out.dtype), stream_ptr()),
            "aggf_gb_regmat",
        )
    return out


def gb_apply(Fg, Pg, cg, sizes, n_id: int, n_ch: int, centers, width: float, clip: float, coef: torch.Tensor):
    l = lib()
    T, G, _ = Fg.shape
    n_cg = cg.shape[1]
    out = torch.empty((T, n_cg, 3), dtype=torch.float64, device=Fg.device)
    if T == 0:
        return out
    with _timed("gb_apply"):
        check(
            l.aggf_gb_apply(ptr(Fg), dtype_code(Fg.dtype), ptr(Pg), ptr(cg), _gb_dtype(Pg, cg, centers), T, G, n_cg,
                            ptr(sizes), n_id, n_ch,
                            ptr(centers), centers.numel(), float(width), float(clip), ptr(coef), coef.shape[1],
                            ptr(out), stream_ptr()),
            "aggf_gb_apply",
        )
    return out


def gb_compact_coefficients(coef: np.ndarray, n_id: int, device):
    """(coef_id (n_cg, n_id), col_ptr, col_idx, col_val) on the device for :func:`gb_apply_cols`: the non-zero
    Gaussian coefficients of every site as compressed rows."""
    coef = np.asarray(coef, dtype=np.float64)
    gauss = coef[:, n_id:]
    rows, cols = np.nonzero(gauss)
    ptr_h = np.zeros(coef.shape[0] + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=coef.shape[0]), out=ptr_h[1:])
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return (to(coef[:, :n_id]) if n_id else None, to(ptr_h), to(cols.astype(np.int32)), to(gauss[rows, cols]))


def gb_apply_cols(Fg, Pg, cg, sizes, n_id: int, centers, width: float, clip: float, compact):
    """gb_apply from the compact coefficient list of :func:`gb_compact_coefficients`; see aggf_gb_apply_cols."""
    coef_id, col_ptr, col_idx, col_val = compact
    T, G, _ = Fg.shape
    n_cg = cg.shape[1]
    out = torch.empty((T, n_cg, 3), dtype=torch.float64, device=Fg.device)
    if T == 0:
        return out
    with _timed("gb_apply"):
        check(
            lib().aggf_gb_apply_cols(ptr(Fg), dtype_code(Fg.dtype), ptr(Pg), ptr(cg), _gb_dtype(Pg, cg, centers), T, G,
                                     n_cg, ptr(sizes), n_id, ptr(coef_id), ptr(col_ptr), ptr(col_idx), ptr(col_val),
                                     ptr(centers), centers.numel(), float(width), float(clip), ptr(out), stream_ptr()),
            "aggf_gb_apply_cols",
        )
    return out


# ------------------------------------------------------------------ K3c / K4b / K4c per-frame contractions


def trjdot_frames(points: torch.Tensor, factor: torch.Tensor, trans: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[t,c,d] = sum_f factor[t,c,f] points[t,f,d] (+ trans[t,c,d]); see aggf_trjdot_frames."""
    T, N, D = points.shape
    if D != 3 or factor.dim() != 3 or factor.shape[0] != T or factor.shape[2] != N:
        raise ValueError(f"shape mismatch: points {tuple(points.shape)}, factor {tuple(factor.shape)}")
    n_cg = factor.shape[1]
    out_dtype = torch.promote_types(points.dtype, factor.dtype)
    if trans is not None:
        if tuple(trans.shape) != (T, n_cg, 3):
            raise ValueError(f"shape mismatch: trans {tuple(trans.shape)}, expected {(T, n_cg, 3)}")
        out_dtype = torch.promote_types(out_dtype, trans.dtype)
        if out_dtype == torch.float64 and points.dtype == factor.dtype == torch.float32:
            points = points.to(torch.float64)  # the kernel's out dtype is the promoted input dtype
        trans = trans.to(out_dtype).contiguous()
    out = torch.empty((T, n_cg, 3), dtype=out_dtype, device=points.device)
    if T == 0 or n_cg == 0:
        return out
    with _timed("trjdot_frames"):
        check(lib().aggf_trjdot_frames(ptr(points), dtype_code(points.dtype), ptr(factor), dtype_code(factor.dtype), T, N,
                                       n_cg, ptr(trans), ptr(out), dtype_code(out_dtype), stream_ptr()),
              "aggf_trjdot_frames")
    return out


def feat_contract(forces: torch.Tensor, feat: torch.Tensor, div: Optional[torch.Tensor], alpha: float,
                  ld: Optional[int] = None) -> torch.Tensor:
    """out (T, ld, 3): sum_a feat[t,a,f] F[t,a,d] + alpha div[t,f,d]; see aggf_feat_contract."""
    T, N, _ = forces.shape
    if feat.dim() != 3 or feat.shape[0] != T or feat.shape[1] != N:
        raise ValueError(f"shape mismatch: forces {tuple(forces.shape)}, feats {tuple(feat.shape)}")
    n_feat = feat.shape[2]
    if div is not None:
        if tuple(div.shape) != (T, n_feat, 3):
            raise ValueError(f"shape mismatch: feats {tuple(feat.shape)}, divs {tuple(div.shape)}")
        xdt = torch.promote_types(feat.dtype, div.dtype)
        feat, div = feat.to(xdt).contiguous(), div.to(xdt).contiguous()
    ld = n_feat if ld is None else int(ld)
    out_dtype = torch.promote_types(forces.dtype, feat.dtype)
    out = torch.empty((T, ld, 3), dtype=out_dtype, device=forces.device)
    if T == 0:
        return out
    with _timed("feat_contract"):
        check(lib().aggf_feat_contract(ptr(forces), dtype_code(forces.dtype), ptr(feat), ptr(div), dtype_code(feat.dtype),
                                       float(alpha), T, N, n_feat, ld, ptr(out), dtype_code(out_dtype), stream_ptr()),
              "aggf_feat_contract")
    return out


def feat_constraint_rows(feat: torch.Tensor, frame_idx: np.ndarray, M: torch.Tensor, site: int):
    """(A (S*n_cg, n_feat), b (S*n_cg, 1)) float64 for one cg site; see aggf_feat_constraint_rows."""
    T, N, n_feat = feat.shape
    idx = np.asarray(frame_idx, dtype=np.int64).reshape(-1)
    if idx.size == 0 or idx.min() < -T or idx.max() >= T:
        raise IndexError("constraint frame index outside the trajectory")
    idx = np.where(idx < 0, idx + T, idx)
    n_cg = M.shape[0]
    S = idx.size
    idx_dev = torch.from_numpy(idx).to(feat.device)
    A = torch.empty((S * n_cg, n_feat), dtype=torch.float64, device=feat.device)
    b = torch.empty((S * n_cg, 1), dtype=torch.float64, device=feat.device)
    check(lib().aggf_feat_constraint_rows(ptr(feat), dtype_code(feat.dtype), T, N, n_feat, ptr(idx_dev), S, ptr(M), n_cg,
                                          int(site), ptr(A), ptr(b), stream_ptr()), "aggf_feat_constraint_rows")
    return A, b


def gb_constraint_rows(Mg: torch.Tensor, gauss: Optional[torch.Tensor], S: int, n_id: int, n_ch: int, n_basis: int,
                       site: int, out_A: Optional[torch.Tensor] = None, out_b: Optional[torch.Tensor] = None,
                       cols: Optional[torch.Tensor] = None):
    """Constraint rows of the fused [id | gb] features; see aggf_gb_constraint_rows.  ``cols`` (int32 device
    array of kept Gaussian columns ch*n_basis + k) gives the compacted layout; ``out_A`` may be wider than
    the feature count (trailing columns are zeroed)."""
    n_cg, G = Mg.shape
    n_cols = n_ch * n_basis if cols is None else int(cols.numel())
    n_feat = n_id + n_cols
    A = out_A if out_A is not None else torch.empty((S * n_cg, n_feat), dtype=torch.float64, device=Mg.device)
    b = out_b if out_b is not None else torch.empty((S * n_cg, 1), dtype=torch.float64, device=Mg.device)
    ld = A.shape[-1]
    gd = _lib.F32 if gauss is None else dtype_code(gauss.dtype)
    check(lib().aggf_gb_constraint_rows(ptr(Mg), ptr(gauss), gd, S, n_cg, G, n_id, n_ch, n_basis, ptr(cols), n_cols, ld,
                                        int(site), ptr(A), ptr(b), stream_ptr()), "aggf_gb_constraint_rows")
    return A, b


def gb_group_overlap(Mg: torch.Tensor) -> torch.Tensor:
    """Mg' Mg (G, G) float64; see aggf_gb_group_overlap."""
    n_cg, G = Mg.shape
    M2 = torch.empty((G, G), dtype=torch.float64, device=Mg.device)
    check(lib().aggf_gb_group_overlap(ptr(Mg), n_cg, G, ptr(M2), stream_ptr()), "aggf_gb_group_overlap")
    return M2


def gb_constraint_gram(M2: torch.Tensor, gauss: Optional[torch.Tensor], S: int, n_id: int, n_ch: int, n_basis: int,
                       out: torch.Tensor, cols: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A'A of gb_constraint_rows' rows into the lower triangle of ``out`` (ld, ld); see aggf_gb_constraint_gram."""
    G = M2.shape[0]
    n_cols = n_ch * n_basis if cols is None else int(cols.numel())
    if out.dim() != 2 or out.shape[0] != out.shape[1] or out.dtype != torch.float64 or not out.is_contiguous():
        raise ValueError("out must be a contiguous square float64 matrix")
    gd = _lib.F32 if gauss is None else dtype_code(gauss.dtype)
    check(lib().aggf_gb_constraint_gram(ptr(M2), ptr(gauss), gd, S, G, n_id, n_ch, n_basis, ptr(cols), n_cols,
                                        out.shape[0], ptr(out), stream_ptr()), "aggf_gb_constraint_gram")
    return out


def gb_distance_range(Pg: torch.Tensor, cg: torch.Tensor, n_ch: int):
    """(rmin, rmax) (n_cg, G) float32: range over frames of every channel's distance to every cg site
    (evaluated in float32 whatever the feature dtype: the caller applies a safety margin)."""
    if Pg.dtype != torch.float32:
        Pg, cg = Pg.to(torch.float32), cg.to(torch.float32)
    T, G, _ = Pg.shape
    n_cg = cg.shape[1]
    rmin = torch.full((n_cg, G), float("inf"), dtype=torch.float32, device=Pg.device)
    rmax = torch.zeros((n_cg, G), dtype=torch.float32, device=Pg.device)
    if T and n_ch:
        check(lib().aggf_gb_distance_range(ptr(Pg), ptr(cg), T, G, n_cg, n_ch, ptr(rmin), ptr(rmax), stream_ptr()),
              "aggf_gb_distance_range")
    return rmin, rmax


def gb_regmat_cols(Fg, Pg, cg, site: int, sizes, n_id: int, cols: torch.Tensor, centers, width: float, clip: float,
                   kbt: float, out: torch.Tensor) -> torch.Tensor:
    """Compact regression matrix of one cg site (listed Gaussian columns only); see aggf_gb_regmat_cols."""
    T, G, _ = Fg.shape
    with _timed("gb_regmat"):
        check(
            lib().aggf_gb_regmat_cols(ptr(Fg), dtype_code(Fg.dtype), ptr(Pg), ptr(cg), _gb_dtype(Pg, cg, centers), T, G,
                                      cg.shape[1], site,
                                      ptr(sizes), n_id, ptr(cols), int(cols.numel()), ptr(centers), centers.numel(),
                                      float(width), float(clip), float(kbt), out.shape[1], ptr(out),
                                      dtype_code(out.dtype), stream_ptr()),
            "aggf_gb_regmat_cols",
        )
    return out


def feat_weights(feat: torch.Tensor, coef: torch.Tensor, out: torch.Tensor, site: int) -> None:
    """out[:, site, :] (T, n_cg, N float64) = feat (T, N, n_feat) . coef (n_feat); see aggf_feat_weights."""
    T, N, n_feat = feat.shape
    n_cg = out.shape[1]
    if T == 0:
        return
    view = out[:, site, :]
    check(lib().aggf_feat_weights(ptr(feat), dtype_code(feat.dtype), T, N, n_feat, ptr(coef), n_cg * N, view.data_ptr(),
                                  stream_ptr()), "aggf_feat_weights")


# ------------------------------------------------------------------ K6 pair-distance variance


def pair_dist_var(x: torch.Tensor) -> torch.Tensor:
    """(N, N) float64 population variance over frames of every pair distance; see aggf_pair_dist_var."""
    l = lib()
    T, N, _ = x.shape
    var = torch.empty((N, N), dtype=torch.float64, device=x.device)
    need = l.aggf_pair_dist_var_workspace_bytes(T, N)
    ws = workspace(need, x.device, "pairs")
    with _timed("pair_var"):
        check(l.aggf_pair_dist_var(ptr(x), T, N, dtype_code(x.dtype), ptr(var), ptr(ws), need, stream_ptr()),
              "aggf_pair_dist_var")
    return var


def pair_dist_moments(x: torch.Tensor):
    """(mean, var) (N, N) float64 of every pair distance over the frames; see aggf_pair_dist_moments."""
    l = lib()
    T, N, _ = x.shape
    mean = torch.empty((N, N), dtype=torch.float64, device=x.device)
    var = torch.empty((N, N), dtype=torch.float64, device=x.device)
    need = l.aggf_pair_dist_var_workspace_bytes(T, N)
    ws = workspace(need, x.device, "pairs")
    with _timed("pair_var"):
        check(l.aggf_pair_dist_moments(ptr(x), T, N, dtype_code(x.dtype), ptr(mean), ptr(var), ptr(ws), need, stream_ptr()),
              "aggf_pair_dist_moments")
    return mean, var


def pair_pool_term(var_r: torch.Tensor, mean_r: torch.Tensor, mean: torch.Tensor, weight: float) -> torch.Tensor:
    """weight * (var_r + (mean_r - mean)^2), in place of var_r; see aggf_pair_pool_term."""
    check(lib().aggf_pair_pool_term(ptr(var_r), ptr(mean_r), ptr(mean), float(weight), var_r.numel(), ptr(var_r),
                                    stream_ptr()), "aggf_pair_pool_term")
    return var_r
