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
        return self.tmap(AugmentedTrajectory.from_trajectory(t=t, kbt=self.kbt, augmenter=self.augmenter))

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
        return Trajectory(coords=t.coords, forces=self.fill_value * t.coords)

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
