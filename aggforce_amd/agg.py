"""Entry point: optimise a force map and apply it (reference: agg.py).

``project_forces`` keeps the reference's signature, ``method=`` plug-in contract
(``method(traj=, coord_map=, constraints=, **kwargs) -> TMap``, agg.py:121-126) and result
keys (agg.py:42-46).  Input arrays may be NumPy arrays (uploaded to the GPU once, results
returned as NumPy arrays) or torch ROCm tensors (everything stays on the device).
"""
from gc import collect
from itertools import product
from typing import Any, Callable, Collection, Dict, Final, List, Mapping, NamedTuple, Tuple, TypeVar, Union

import numpy as np

from . import _kernels as K
from .constraints import Constraints, guess_pairwise_constraints
from .distributed import (all_reduce_sum_, all_reduce_sum_sym_, cancel_overlap, overlap_with_next_collective,
                          world_size)
from .map import LinearMap, SeperableTMap, TMap
from .qp import qp_linear_map
from .trajectory import Trajectory

PROJECT_FORCES_CNSTR_AUTO: Final = "auto"

SCORES_KNAME: Final = "scores"
SDS_KNAME: Final = "sds"
NRUNS_KNAME: Final = "n_runs"

PROJFORCES_KNAME: Final = "mapped_forces"
PROJCOORDS_KNAME: Final = "mapped_coords"
TMAP_KNAME: Final = "tmap"
RESIDUAL_KNAME: Final = "residual"
CONSTRAINTS_KNAME: Final = "constraints"

T = TypeVar("T")


def force_smoothness(array, comm=None) -> float:
    """Mean squared element (reference agg.py:291-297), reduced on the GPU in a fixed order.

    With ``comm`` the sum of squares and the element count are summed over the ranks.
    """
    import torch

    x = K.as_device(array)
    if world_size(comm) == 1:  # one device-to-host scalar, no staging of the count on the device
        n = x.numel()
        return float(K.sumsq(x).item()) / n if n else float("nan")
    acc = torch.cat([K.sumsq(x), torch.tensor([float(x.numel())], dtype=torch.float64, device=x.device)])
    all_reduce_sum_(acc, comm)
    s, n = acc.tolist()
    return float(s / n) if n else float("nan")


def _mean_square(sumsq, count: int, comm=None) -> float:
    """mean of squares from a device sum and an element count (summed over the ranks with ``comm``)."""
    import torch

    if world_size(comm) == 1:
        return float(sumsq.item()) / count if count else float("nan")
    acc = torch.cat([sumsq.reshape(1).to(torch.float64),
                     torch.tensor([float(count)], dtype=torch.float64, device=sumsq.device)])
    all_reduce_sum_(acc, comm)
    s, n = acc.tolist()
    return float(s / n) if n else float("nan")


def project_forces(
    coords,
    forces,
    coord_map: LinearMap,
    constrained_inds: Union[Constraints, str, None] = PROJECT_FORCES_CNSTR_AUTO,
    method: Callable[..., TMap] = qp_linear_map,
    **kwargs,
) -> Dict[str, Any]:
    """Produce an optimised force map and the mapped trajectory (reference agg.py:49-136).

    coords, forces: (n_steps, n_sites, 3).  ``constrained_inds``: set of frozensets of
    constrained sites, or "auto" to guess them from ``coords`` with
    ``guess_pairwise_constraints`` (all frames are passed, as in the reference).  ``method``
    is called as ``method(traj=, coord_map=, constraints=, **kwargs)`` and must return a TMap.
    A ``comm=`` keyword (torch.distributed group over which the frames are sharded) is
    forwarded to ``method`` and also used to reduce the residual.

    Returns a dict with keys mapped_coords, mapped_forces, tmap, residual, constraints.
    """
    if isinstance(constrained_inds, str) and constrained_inds == PROJECT_FORCES_CNSTR_AUTO:
        if coords is None:
            raise ValueError(
                f"If constrained_inds is {PROJECT_FORCES_CNSTR_AUTO}, coords cannot be None."
            )
        # (with comm= the per-rank distance statistics are pooled exactly: every rank gets the same set)
        constrained_inds = guess_pairwise_constraints(coords, comm=kwargs.get("comm"))
    with K.upload_cache():
        t = Trajectory(coords=coords, forces=forces)
        fused_ss = None  # sum of squares of the mapped forces when the apply kernel accumulated it
        # Frames sharded over ranks: the coordinate gather of a slice map starts when the fit reaches its Gram
        # all-reduce (link-bound, compute units idle) and runs beside it on the side stream.
        early = {}
        hook = None
        if world_size(kwargs.get("comm")) > 1 and isinstance(coord_map, LinearMap) and method is qp_linear_map:
            def hook():
                early["pending"] = coord_map.map_async(t.coords)

            overlap_with_next_collective(hook)
        try:
            traj_map: TMap = method(traj=t, coord_map=coord_map, constraints=constrained_inds, **kwargs)
        except BaseException:
            if early.get("pending") is not None:
                early["pending"].discard()
            raise
        finally:
            if hook is not None:
                cancel_overlap(hook)
        pending = early.get("pending")
        if pending is not None and not (type(traj_map) is SeperableTMap and traj_map.coord_map is coord_map):
            pending.discard()
            pending = None
        elif (pending is None and type(traj_map) is SeperableTMap and traj_map.coord_map is coord_map
              and isinstance(coord_map, LinearMap)):
            # a slice map's gather (HBM-bound) goes to a side stream underneath the force apply (MFMA-bound):
            # c3 869 ms/step against 874-878 one after the other; started before the fit it slows the
            # Gram kernel by as much as it saves (872-876), underneath the solve it doubles the solve (874-876)
            pending = coord_map.map_async(t.coords)
        if pending is not None:
            try:
                fmap = traj_map.force_map
                if isinstance(fmap, LinearMap):
                    mapped_forces, fused_ss = fmap.call_with_sumsq(t.forces)
                else:
                    mapped_forces = fmap(t.forces)
                mapped_coords = pending.result()
                pending = None
            finally:
                if pending is not None:
                    pending.discard()
        else:
            mapped = traj_map(t)
            mapped_coords, mapped_forces = mapped.coords, mapped.forces
        if fused_ss is not None:
            residual = _mean_square(fused_ss, int(np.prod(mapped_forces.shape)), kwargs.get("comm"))
        else:
            residual = force_smoothness(mapped_forces, comm=kwargs.get("comm"))
    return {
        PROJCOORDS_KNAME: mapped_coords,
        PROJFORCES_KNAME: mapped_forces,
        TMAP_KNAME: traj_map,
        RESIDUAL_KNAME: residual,
        CONSTRAINTS_KNAME: constrained_inds,
    }


_GRAM_REUSE_GRID_ARGS: Final = frozenset({"l2_regularization"})
_GRAM_REUSE_FIXED_ARGS: Final = frozenset(
    {"coord_map", "constrained_inds", "method", "l2_regularization", "solver_args", "gram_dtype", "comm"}
)


def _take_frames(arr, idx):
    if hasattr(arr, "detach"):
        if arr.is_cuda and arr.dtype in (K.torch_dtype(np.float32), K.torch_dtype(np.float64)):
            return K.take_frames(arr, idx)  # one gather kernel of the library, not an ATen index op
        import torch

        return arr[torch.as_tensor(idx, device=arr.device)]
    return arr[idx]


def _gram_reuse_applicable(grid_names, kwargs) -> bool:
    """The one-pass cross-validation covers the linear optimiser with explicit constraints."""
    if kwargs.get("method", qp_linear_map) is not qp_linear_map or "coord_map" not in kwargs:
        return False
    if not set(grid_names) <= _GRAM_REUSE_GRID_ARGS or not set(kwargs) <= _GRAM_REUSE_FIXED_ARGS:
        return False
    cons = kwargs.get("constrained_inds", PROJECT_FORCES_CNSTR_AUTO)
    return not isinstance(cons, str)  # "auto" guesses constraints per training subset


def _grid_cv_gram_reuse(grid, forces, folds, kwargs) -> Dict[str, Dict[Any, Any]]:
    """Cross-validation of the linear map in ONE pass over the frames.

    G = sum over frames is additive, so each fold's Gram is formed once (K1), the training Gram of
    fold k is ``total - G_k`` and the hold-out score of the trained reduced coefficients X is
    ``sum_i x_i' G_k x_i / (3 T_k n_cg)`` -- no second pass over training or validation frames
    (replaces the loop body of the reference's agg.py:208-231).
    """
    import torch
    from .qp.qplinear import LinearProblem

    comm = kwargs.get("comm")
    f_dev = forces
    prob = LinearProblem(kwargs["coord_map"], kwargs.get("constrained_inds"), f_dev.device)
    fold_grams = torch.stack([prob.gram(_take_frames(f_dev, idx).contiguous(), kwargs.get("gram_dtype")) for idx in folds])
    counts = torch.tensor([float(len(idx)) for idx in folds], dtype=torch.float64, device=f_dev.device)
    all_reduce_sum_sym_(fold_grams, comm)
    all_reduce_sum_(counts, comm)
    return _score_folds(grid, fold_grams, counts.tolist(), prob, kwargs)


def _score_folds(grid, fold_grams, counts, prob, kwargs) -> Dict[str, Dict[Any, Any]]:
    """Grid points x folds from per-fold Gram matrices: training matrix ``total - G_k``, one solve, hold-out score as
    the quadratic form of the reduced coefficients in ``G_k``."""
    import torch

    n_folds = fold_grams.shape[0]
    total = fold_grams[0].clone()
    for k in range(1, n_folds):
        K.axpby(1.0, total, 1.0, fold_grams[k], out=total)
    n_cg = prob.A.shape[0]
    results: Dict[str, Dict[Any, Any]] = {SCORES_KNAME: {}, SDS_KNAME: {}, NRUNS_KNAME: {}}
    train = torch.empty_like(total)
    for label, args in grid:
        l2 = dict(kwargs, **args).get("l2_regularization", 0.0)
        scores = []
        for k in range(n_folds):
            try:
                K.axpby(1.0, total, -1.0, fold_grams[k], out=train)
                X = prob.solve(train, l2)
                q = K.gram_quadform(fold_grams[k], X)
                scores.append(float(q.cpu().numpy().sum()) / (3.0 * counts[k] * n_cg))
            except ValueError as e:
                print(e)
        results[SCORES_KNAME][label] = mean(scores)
        results[SDS_KNAME][label] = sample_sd(scores)
        results[NRUNS_KNAME][label] = len(scores)
    return results


_NOISED_REUSE_FIXED_ARGS: Final = frozenset(
    {"coord_map", "constrained_inds", "method", "var", "kbt", "seed", "l2_regularization", "solver_args", "gram_dtype",
     "comm"}
)


def _noised_reuse_applicable(grid_names, kwargs) -> bool:
    """joptgauss_map with explicit constraints and a grid over ``l2_regularization`` only: one noise realisation, one
    pass over the frames (``qp/gauss.py:cv_joptgauss_fold_grams``); layouts the in-place fit does not take return
    None there and the loop is followed."""
    from .qp import joptgauss_map

    if kwargs.get("method") is not joptgauss_map or not {"coord_map", "var", "kbt"} <= set(kwargs):
        return False
    if not set(grid_names) <= _GRAM_REUSE_GRID_ARGS or not set(kwargs) <= _NOISED_REUSE_FIXED_ARGS:
        return False
    return not isinstance(kwargs.get("constrained_inds", PROJECT_FORCES_CNSTR_AUTO), str)


_FEAT_REUSE_FIXED_ARGS: Final = frozenset(
    {"coord_map", "constrained_inds", "method", "featurizer", "kbt", "n_constraint_frames", "l2_regularization"}
)


def _feat_reuse(grid, grid_names, kwargs):
    """The featuriser's one-pass cross-validation routine if it covers this call, else None: ``qp_feat_linear_map``
    with the fused [id_feat | gb_feat] featurisers, explicit constraints, a grid over ``l2_regularization`` only with
    every value > 0 (columns that vanish on a training subset keep an exactly zero coefficient only then), no ``comm``."""
    from .qp import qp_feat_linear_map

    if kwargs.get("method") is not qp_feat_linear_map or not {"coord_map", "featurizer", "kbt"} <= set(kwargs):
        return None
    if not set(grid_names) <= _GRAM_REUSE_GRID_ARGS or not set(kwargs) <= _FEAT_REUSE_FIXED_ARGS:
        return None
    if isinstance(kwargs.get("constrained_inds", PROJECT_FORCES_CNSTR_AUTO), str):
        return None
    if not all(float(dict(kwargs, **args).get("l2_regularization", 1e1)) > 0.0 for _, args in grid):
        return None
    return getattr(kwargs["featurizer"], "fused_cv", None)


def _grid_cv_feat_reuse(cv, grid, coords, forces, folds, kwargs, method_rng=None) -> Union[Dict[str, Dict[Any, Any]], None]:
    l2_values = [float(dict(kwargs, **args).get("l2_regularization", 1e1)) for _, args in grid]
    cons = kwargs.get("constrained_inds")
    table = cv(coords, forces, kwargs["coord_map"], kwargs["kbt"], kwargs.get("n_constraint_frames", 20),
               set() if cons is None else cons, l2_values, folds, method_rng)
    if table is None:  # does not fit the device: the loop
        return None
    results: Dict[str, Dict[Any, Any]] = {SCORES_KNAME: {}, SDS_KNAME: {}, NRUNS_KNAME: {}}
    for (label, _), row in zip(grid, table):
        scores = [v for v in row if v is not None]
        results[SCORES_KNAME][label] = mean(scores)
        results[SDS_KNAME][label] = sample_sd(scores)
        results[NRUNS_KNAME][label] = len(scores)
    return results


def project_forces_grid_cv(
    cv_arg_dict: Mapping[str, List[T]],
    coords,
    forces,
    n_folds: int = 5,
    rng=None,
    reuse_gram: bool = True,
    method_rng=None,
    cv_noise=None,
    **kwargs,
) -> Dict[str, Dict[NamedTuple, T]]:
    """Grid cross-validation over project_forces arguments (reference agg.py:142-235).

    For every grid point the map is trained on the frames outside each fold and scored by
    ``force_smoothness`` of the mapped hold-out forces; returns ``{"scores", "sds", "n_runs"}``
    keyed by the grid point.  ``rng`` (a numpy Generator) makes the fold shuffle reproducible; the
    reference uses an unseeded generator.  ``method_rng`` is handed to ``method`` as ``rng=`` in every
    fit (methods that sample, e.g. the constraint frames of ``qp_feat_linear_map``).  ``joptgauss_map``
    with explicit constraints and a grid over ``l2_regularization`` also takes a one-pass form: ONE
    noise realisation for all frames (the loop draws afresh for every fit and application), per-fold
    Gram matrices of the extended system; ``cv_noise`` (n_frames, n_cg, 3) standard normals fixes that
    realisation (tests).  When the grid runs over ``l2_regularization`` of the
    linear optimiser with explicit constraints and ``reuse_gram`` is true, all folds and grid
    points share one pass over the frames (``_grid_cv_gram_reuse``); the same holds for
    ``qp_feat_linear_map`` with the fused ``[id_feat | gb_feat]`` featurisers and every
    ``l2_regularization`` > 0 (``_grid_cv_feat_reuse``); otherwise the reference's loop over
    ``project_forces`` calls is followed.

    (The reference calls ``trained_tmap.from_arrays`` at agg.py:224, which no TMap defines; the
    intended ``map_arrays`` is used here.)
    """
    n_frames = forces.shape[0]
    frames = np.arange(n_frames)
    (np.random.default_rng() if rng is None else rng).shuffle(frames)
    folds = np.array_split(frames, n_folds)
    grid = process_cvargs(cv_arg_dict)
    if reuse_gram and _gram_reuse_applicable(list(cv_arg_dict.keys()), kwargs):
        f_dev = K.as_device(forces)
        if not K.has_nan(f_dev):  # NaN handling follows the generic path
            return _grid_cv_gram_reuse(grid, f_dev, folds, kwargs)
        del f_dev
    if reuse_gram and _noised_reuse_applicable(list(cv_arg_dict.keys()), kwargs):
        from .qp.gauss import cv_joptgauss_fold_grams

        from .distributed import agree_on_min

        f_dev, c_dev = K.as_device(forces), K.as_device(coords)
        comm = kwargs.get("comm")
        made = None
        clean = not K.has_nan(f_dev) and not K.has_nan(c_dev)
        if agree_on_min(int(clean), comm, f_dev.device):  # (a NaN on any rank sends every rank to the loop)
            made = cv_joptgauss_fold_grams(c_dev, f_dev, kwargs["coord_map"], kwargs["var"], kwargs["kbt"],
                                           kwargs.get("constrained_inds"), kwargs.get("seed"), folds,
                                           kwargs.get("gram_dtype"), noise=cv_noise, comm=comm)
        if made is not None:
            import torch

            counts = torch.tensor([float(len(f)) for f in folds], dtype=torch.float64, device=f_dev.device)
            all_reduce_sum_(counts, comm)
            del f_dev, c_dev
            return _score_folds(grid, made[0], counts.tolist(), made[1], kwargs)
        del f_dev, c_dev
    feat_cv = _feat_reuse(grid, list(cv_arg_dict.keys()), kwargs) if reuse_gram else None
    if feat_cv is not None and not K.has_nan(K.as_device(forces)) and not K.has_nan(K.as_device(coords)):
        done = _grid_cv_feat_reuse(feat_cv, grid, coords, forces, folds, kwargs, method_rng)
        if done is not None:
            return done
    results: Dict[str, Dict[Any, Any]] = {SCORES_KNAME: {}, SDS_KNAME: {}, NRUNS_KNAME: {}}
    take = _take_frames
    for label, args in grid:
        scores = []
        merged = dict(kwargs, **args)
        if method_rng is not None:
            merged["rng"] = method_rng
        for k, val_idx in enumerate(folds):
            train_idx = np.concatenate([f for j, f in enumerate(folds) if j != k])
            try:
                tmap = project_forces(coords=take(coords, train_idx), forces=take(forces, train_idx), **merged)[
                    TMAP_KNAME
                ]
                _, val_forces = tmap.map_arrays(take(coords, val_idx), take(forces, val_idx))
                scores.append(force_smoothness(val_forces, comm=merged.get("comm")))
                del tmap
            except ValueError as e:
                print(e)
            collect()
        results[SCORES_KNAME][label] = mean(scores)
        results[SDS_KNAME][label] = sample_sd(scores)
        results[NRUNS_KNAME][label] = len(scores)
    return results


def process_cvargs(arg_dict: Mapping[str, List[Any]]) -> List[Tuple[NamedTuple, Dict[str, Any]]]:
    """Cartesian grid of argument values as (namedtuple key, kwargs dict) pairs (agg.py:238-288)."""
    names = list(arg_dict.keys())
    CVArgs = NamedTuple("CVArgs", [(n, Any) for n in names])  # type: ignore [misc]
    grid = []
    for values in product(*[arg_dict[n] for n in names]):
        grid.append((CVArgs(*values), dict(zip(names, values))))
    return grid


def mean(s: Collection[float]) -> Union[float, None]:
    """Arithmetic mean, None for an empty collection."""
    return None if len(s) == 0 else sum(s) / len(s)


def sample_sd(s: Collection[float]) -> Union[float, None]:
    """Sample standard deviation, None for an empty collection."""
    m = mean(s)
    if m is None:
        return None
    return (sum((o - m) ** 2 for o in s) / (len(s) - 1)) ** 0.5
