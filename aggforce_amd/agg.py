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
from .distributed import all_reduce_sum_
from .map import LinearMap, TMap
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
    acc = torch.cat([K.sumsq(x), torch.tensor([float(x.numel())], dtype=torch.float64, device=x.device)])
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
        constrained_inds = guess_pairwise_constraints(coords)
    with K.upload_cache():
        t = Trajectory(coords=coords, forces=forces)
        traj_map: TMap = method(traj=t, coord_map=coord_map, constraints=constrained_inds, **kwargs)
        mapped = traj_map(t)
        residual = force_smoothness(mapped.forces, comm=kwargs.get("comm"))
    return {
        PROJCOORDS_KNAME: mapped.coords,
        PROJFORCES_KNAME: mapped.forces,
        TMAP_KNAME: traj_map,
        RESIDUAL_KNAME: residual,
        CONSTRAINTS_KNAME: constrained_inds,
    }


def project_forces_grid_cv(
    cv_arg_dict: Mapping[str, List[T]],
    coords,
    forces,
    n_folds: int = 5,
    rng=None,
    **kwargs,
) -> Dict[str, Dict[NamedTuple, T]]:
    """Grid cross-validation over project_forces arguments (reference agg.py:142-235).

    Host-side hyper-parameter loop around the hot path.  ``rng`` (a numpy Generator) makes the
    fold shuffle reproducible; the reference uses an unseeded generator.
    """
    n_frames = forces.shape[0]
    frames = np.arange(n_frames)
    (np.random.default_rng() if rng is None else rng).shuffle(frames)
    folds = np.array_split(frames, n_folds)
    results: Dict[str, Dict[Any, Any]] = {SCORES_KNAME: {}, SDS_KNAME: {}, NRUNS_KNAME: {}}

    def take(arr, idx):
        if hasattr(arr, "detach"):
            import torch

            return arr[torch.as_tensor(idx, device=arr.device)]
        return arr[idx]

    for label, args in process_cvargs(cv_arg_dict):
        scores = []
        merged = dict(kwargs, **args)
        for k, val_idx in enumerate(folds):
            train_idx = np.concatenate([f for j, f in enumerate(folds) if j != k])
            try:
                tmap = project_forces(coords=take(coords, train_idx), forces=take(forces, train_idx), **merged)[
                    TMAP_KNAME
                ]
                _, val_forces = tmap.map_arrays(take(coords, val_idx), take(forces, val_idx))
                scores.append(force_smoothness(val_forces))
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
