"""Trajectory-level maps (reference: map/tmap.py).

A TMap maps a Trajectory (coordinates and forces together).  ``SeperableTMap`` applies one
map to the coordinates and one to the forces; ``CLAFTMap`` pairs a LinearMap for
coordinates with a configuration-dependent CLAMap for forces; ``AugmentedTMap`` first
extends the trajectory with noise sites.  ``ComposedTMap``, ``NullForcesTMap`` and
``RATMap`` complete the reference's set (tmap.py:258-437).
"""
from abc import ABC, abstractmethod
from typing import Callable, Final, Iterable, Tuple, TypeVar
from warnings import warn

import numpy as np

from ..trajectory.augment import Augmenter
from ..trajectory.core import (
    AugmentedTrajectory,
    CoordsTrajectory,
    ForcesTrajectory,
    Trajectory,
    _concat_sites,
)
from .core import CLAMap

ArrayTransform = Callable
_T = TypeVar("_T", bound="TMap")


class TMap(ABC):
    """Interface: ``__call__(Trajectory) -> Trajectory``, ``map_arrays``, ``astype``."""

    @abstractmethod
    def __init__(self) -> None:
        """Initialize."""

    @abstractmethod
    def __call__(self, t: Trajectory) -> Trajectory:
        """Map a Trajectory to a new instance."""

    def map_arrays(self, coords, forces) -> Tuple:
        """Map coordinate and force arrays of shape (n_frames, n_sites, n_dims)."""
        mapped = self(Trajectory(coords=coords, forces=forces))
        return (mapped.coords, mapped.forces)

    @abstractmethod
    def astype(self: _T, *args, **kwargs) -> _T:
        """Same map at another numerical precision."""


def _astype_pair(obj, *args, **kwargs):
    try:
        return obj.__class__(
            coord_map=obj.coord_map.astype(*args, **kwargs),
            force_map=obj.force_map.astype(*args, **kwargs),
        )
    except AttributeError as e:
        raise TypeError("Underlying coord_map and/or force_map do not support astype.") from e


class SeperableTMap(TMap):
    """Independent maps for coordinates and for forces (reference tmap.py:85-146)."""

    def __init__(self, coord_map: ArrayTransform, force_map: ArrayTransform) -> None:
        self.coord_map = coord_map
        self.force_map = force_map

    def __call__(self, t: Trajectory) -> Trajectory:
        return Trajectory(coords=self.coord_map(t.coords), forces=self.force_map(t.forces))

    def astype(self, *args, **kwargs) -> "SeperableTMap":
        return _astype_pair(self, *args, **kwargs)


class CLAFTMap(TMap):
    """LinearMap for coordinates, CLAMap (coordinates as copoints) for forces (tmap.py:149-198)."""

    def __init__(self, coord_map: ArrayTransform, force_map: CLAMap) -> None:
        self.coord_map = coord_map
        self.force_map = force_map

    def __call__(self, t: Trajectory) -> Trajectory:
        return Trajectory(
            coords=self.coord_map(t.coords), forces=self.force_map(points=t.forces, copoints=t.coords)
        )

    def astype(self, *args, **kwargs) -> "CLAFTMap":
        return _astype_pair(self, *args, **kwargs)


class AugmentedTMap(TMap):
    """Augment the trajectory with ``augmenter``, then apply ``aug_tmap`` (tmap.py:201-255)."""

    def __init__(self, aug_tmap: TMap, augmenter: Augmenter, kbt: float) -> None:
        self.tmap: Final = aug_tmap
        self.augmenter: Final = augmenter
        self.kbt: Final = kbt

    def __call__(self, t: Trajectory) -> Trajectory:
        fused = self._call_without_extended_arrays(t)
        if fused is not None:
            return fused
        return self.tmap(AugmentedTrajectory.from_trajectory(t=t, kbt=self.kbt, augmenter=self.augmenter))

    def _call_without_extended_arrays(self, t: Trajectory):
        """``self.tmap`` on the extended trajectory [x | y], [F - Fa C | Fa] without forming it, for the map
        joptgauss_map returns (linear force map W over N + n_cg sites, coordinate map = the generated sites):
        W [F - Fa C | Fa] = W_N F + (W_a - W_N C') Fa, mapped coordinates = y.  None when the pieces are of
        another kind (custom augmenters, other TMaps, NaN handling on the input)."""
        from .. import _kernels as K
        from .core import LinearMap

        noise_sites = getattr(self.augmenter, "noise_sites", None)
        sub = self.tmap
        if (noise_sites is None or type(sub) is not SeperableTMap or not isinstance(sub.force_map, LinearMap)
                or not isinstance(sub.coord_map, LinearMap) or isinstance(t, AugmentedTrajectory)):
            return None
        n_real = t.forces.shape[1]
        premap_map = getattr(self.augmenter, "premap_map", None)
        premap = premap_map(n_real) if premap_map is not None else None
        if premap is None or premap.n_fg_sites != n_real:
            return None
        n_aug = premap.n_cg_sites
        W = sub.force_map.standard_matrix
        idx = sub.coord_map._onehot_index()
        if (W.shape[1] != n_real + n_aug or idx is None or len(idx) != n_aug
                or not np.array_equal(idx, np.arange(n_real, n_real + n_aug))):
            return None
        import torch

        forces = K.as_device(t.forces)
        if forces.shape[0] == 0:
            return None
        # W_N F first, with the NaN scan of F fused into the same pass (a separate scan of the 12 GB of BASELINE
        # config 4 cost 3.3 ms per application): a NaN sends the whole call to the general path, whose NaN policy
        # acts on the extended array like the reference's -- no noise has been drawn yet at this point
        W_n = W[:, :n_real]
        w_dev = sub.force_map._device_matrix(torch.float64, forces.device)[:, :n_real].contiguous()
        probe = K.take_flag(forces.device)
        main = K.linearmap_apply(forces, w_dev, nan_probe=probe)
        if K.read_flag(probe):
            return None
        y, fa, cols = noise_sites(t.coords, self.kbt)
        cp, ci, cv = (x.cpu().numpy() for x in cols)
        C = np.zeros((n_aug, n_real))
        C[ci, np.repeat(np.arange(n_real), np.diff(cp))] = cv
        D = np.ascontiguousarray(W[:, n_real:] - W_n @ C.T)
        side = K.linearmap_apply(fa, torch.from_numpy(D).to(forces.device))
        if main.dtype != torch.float64 or side.dtype != torch.float64:
            main, side = main.to(torch.float64), side.to(torch.float64)
        out = K.axpby(1.0, main, 1.0, side, out=main)
        if out.dtype != K.torch_dtype(np.result_type(K.np_dtype_of(t.forces), W.dtype)):
            out = out.to(K.torch_dtype(np.result_type(K.np_dtype_of(t.forces), W.dtype)))
        return Trajectory(coords=K.like_input(y, t.coords), forces=K.like_input(out, t.forces))

    def astype(self, *args, **kwargs) -> "AugmentedTMap":
        return self.__class__(
            aug_tmap=self.tmap.astype(*args, **kwargs),
            augmenter=self.augmenter.astype(*args, **kwargs),
            kbt=self.kbt,
        )


class ComposedTMap(TMap):
    """Composition of TMaps; ``submaps[-1]`` is applied first (tmap.py:258-315)."""

    def __init__(self, submaps: Iterable[TMap]) -> None:
        self.submaps: Final = list(submaps)

    def __call__(self, t: Trajectory) -> Trajectory:
        for tm in reversed(self.submaps):
            t = tm(t)
        return t

    def __getitem__(self, idx: int, /) -> TMap:
        return self.submaps[idx]

    def astype(self, *args, **kwargs) -> "ComposedTMap":
        return self.__class__(submaps=[x.astype(*args, **kwargs) for x in self.submaps])


class NullForcesTMap(TMap):
    """Adds (or overwrites) a null force entry: forces = fill_value * coords (tmap.py:321-405).

    Accepts a CoordsTrajectory or a Trajectory; the default fill is NaN.
    """

    def __init__(self, warn_input_forces: bool = True, fill_value=float("nan")) -> None:
        self.warn_input_forces = warn_input_forces
        self.fill_value = fill_value

    def __call__(self, t: CoordsTrajectory) -> Trajectory:
        if isinstance(t, ForcesTrajectory) and self.warn_input_forces:
            warn("Discarding forces on input trajectory.", stacklevel=0)
        from .. import _kernels as K

        return Trajectory(coords=t.coords, forces=K.scaled(t.coords, self.fill_value))

    def map_arrays(self, coords, forces=None) -> Tuple:
        """Like TMap.map_arrays, but ``forces`` may be omitted."""
        t = CoordsTrajectory(coords=coords) if forces is None else Trajectory(coords=coords, forces=forces)
        derived = self(t)
        return (derived.coords, derived.forces)

    def astype(self, *args, **kwargs) -> "NullForcesTMap":  # noqa: ARG002
        return self.__class__(warn_input_forces=self.warn_input_forces, fill_value=self.fill_value)


class RATMap:
    """Maps only the real sites of an AugmentedTrajectory; augmenting sites are kept (tmap.py:408-437)."""

    def __init__(self, tmap: TMap) -> None:
        self.tmap = tmap

    def __call__(self, t: AugmentedTrajectory) -> Trajectory:
        coords, forces = self.tmap.map_arrays(t.coords[:, t.real_slice, :], t.forces[:, t.real_slice, :])
        return Trajectory(
            coords=_concat_sites([coords, t.coords[:, t.aug_slice, :]]),
            forces=_concat_sites([forces, t.forces[:, t.aug_slice, :]]),
        )
