"""Array primitives and small helpers (reference: util.py).

``trjdot`` is the batched contraction behind every map application; with a 2-D factor
it runs on the GPU (K3, ``aggf_linearmap_apply``).  ``Curry``/``curry`` bind featuriser
options exactly like the reference's helpers (util.py:146-252).
"""
from typing import Any, Callable, Generic, Iterable, List, TypeVar

import numpy as np

T = TypeVar("T")
R = TypeVar("R")


def _to_numpy(x) -> np.ndarray:
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def distances(
    xyz,
    cross_xyz=None,
    return_matrix: bool = True,
    return_displacements: bool = False,
) -> np.ndarray:
    """Per-frame distance matrices (host helper; reference util.py:12-76)."""
    if cross_xyz is not None and not return_matrix:
        raise ValueError("Cross distances only supported when return_matrix is truthy.")
    if return_displacements and not return_matrix:
        raise ValueError("Displacements only supported when return_matrix is truthy.")
    xyz = _to_numpy(xyz)
    other = xyz if cross_xyz is None else _to_numpy(cross_xyz)
    disp = xyz[:, None, :, :] - other[:, :, None, :]
    if return_displacements:
        return disp
    dist = np.sqrt(np.sum(disp * disp, axis=-1))
    if return_matrix:
        return dist
    i0, i1 = np.triu_indices(dist.shape[-1], k=1)
    return dist[:, i0, i1]


def trjdot(points, factor):
    """out[t,c,d] = sum_f factor[c,f] points[t,f,d]  (reference util.py:79-125).

    points: (n_steps, n_sites, 3).  factor: (n_cg, n_sites) -> GPU kernel K3; a 3-D factor
    (n_steps, n_cg, n_sites) (per-frame maps, the reference's "...fd,...cf->...cd" branch) ->
    the streaming kernel K3c (``aggf_trjdot_frames``).
    The result has NumPy's promoted dtype and the container type of ``points``.
    """
    from . import _kernels as K

    fdim = factor.dim() if hasattr(factor, "dim") else np.ndim(factor)
    if fdim not in (2, 3):
        raise ValueError("Factor matrix is an incompatible shape.")
    out_np = np.result_type(K.np_dtype_of(points), K.np_dtype_of(factor))
    if out_np not in (np.float32, np.float64):
        out_np = np.dtype(np.float64)
    out_t = K.torch_dtype(out_np)
    p = K.as_device(points)
    if fdim == 2:
        return K.like_input(K.linearmap_apply(p, K.as_device(factor, out_t)), points)
    return K.like_input(K.trjdot_frames(p, K.as_device(factor)), points)


def flatten(nested_list: Iterable[Iterable[Any]]) -> List[Any]:
    """Flatten one level of nesting."""
    out: List[Any] = []
    for sub in nested_list:
        out.extend(sub)
    return out


def curry(func: Callable[..., T], *args, **kwargs) -> Callable[..., T]:
    """g(*a, **k) = func(*a, *args, **k, **kwargs)  (reference util.py:146-174)."""

    def bound(*a, **k) -> T:
        return func(*a, *args, **k, **kwargs)

    bound.func, bound.args, bound.kwargs = func, args, kwargs  # introspection (fused featurisers)
    return bound


class Curry(Generic[R]):
    """Self-describing callable form of ``curry`` (reference util.py:181-252)."""

    def __init__(self, func: Callable[..., R], *args, **kwargs) -> None:
        self.func = func
        self.args = args
        self.kwargs = kwargs

    def __call__(self, *a, **k) -> R:
        return self.func(*a, *self.args, **k, **self.kwargs)

    def __repr__(self) -> str:
        parts = [f"{self.__class__}():", "C:", repr(self.func)]
        if self.args:
            parts += ["Ar:", repr(self.args)]
        if self.kwargs:
            parts += ["Kw:", repr(self.kwargs)]
        return " ".join(parts)

    def __str__(self) -> str:
        pad = "    "
        lines = [f"{self.__class__} instance:", "callable:"]
        lines += [pad + s for s in str(self.func).split("\n")]
        lines.append("args:")
        lines += [pad + s for s in str(self.args).split("\n")]
        lines.append("kwargs:")
        lines += [pad + s for s in str(self.kwargs).split("\n")]
        return "\n".join(lines)
