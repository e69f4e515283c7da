"""Trajectory containers (reference: trajectory/core.py).

The containers hold ``(n_frames, n_sites, n_dims)`` arrays -- NumPy arrays or torch ROCm
tensors -- and add shape checks, slicing, copying and dtype conversion.
``AugmentedTrajectory`` extends the state space with noise sites produced by an
``Augmenter`` (trajectory/core.py:227-603); with the built-in Gaussian augmenters the
whole augmentation (sampling, log-gradients, force correction, concatenation) is one
fused HIP pass (K5, ``aggf_condnormal_augment``).
"""
from copy import deepcopy
from typing import Any, Callable, NoReturn, Optional, Tuple, TypeVar

import numpy as np

from .augment import Augmenter

A = TypeVar("A")


def _astype(arr, *args, **kwargs):
    if hasattr(arr, "detach"):  # torch tensor: accept a NumPy/torch dtype as first argument
        from .._kernels import torch_dtype
        import torch

        dt = args[0] if args else kwargs.get("dtype")
        return arr.to(dt if isinstance(dt, torch.dtype) else torch_dtype(dt))
    return arr.astype(*args, **kwargs)


def _copy(arr):
    return arr.clone() if hasattr(arr, "clone") else arr.copy()


def _concat_sites(parts):
    if any(hasattr(p, "detach") for p in parts):
        import torch
        from .._kernels import as_device

        from .._kernels import concat_sites

        parts = [as_device(p) for p in parts]
        out = parts[0]
        for p in parts[1:]:
            out = concat_sites(out, p)  # (aggf_concat_sites: conversion to the promoted dtype rides along)
        return out
    return np.concatenate(parts, axis=1)


def _augment_concat(coords, forces, aug_coords, real_corr, aug_lgrad, kbt):
    """([x ; y], [F + kbt d/dx ; kbt d/dy]) of the general Augmenter protocol (reference trajectory/core.py:384-390).
    Dtypes follow NumPy's promotion as in the reference: coordinates come out in promote(x, y), forces in
    promote(F, d/dx, d/dy) -- float64 forces stay float64 beside float32 coordinates.  When the coordinates and forces
    share a dtype (every configuration of BASELINE.json) both arrays leave ONE pass (``aggf_augment_concat``: the two
    scaled sums ride along with the concatenating copy); otherwise the coordinates are concatenated by
    ``aggf_concat_sites`` and the forces by a second call of the fused kernel on the forces alone."""
    import torch

    from .. import _kernels as K

    c, f, y = K.as_device(coords), K.as_device(forces), K.as_device(aug_coords)
    corr, lg = K.as_device(real_corr), K.as_device(aug_lgrad)
    gdt = torch.promote_types(corr.dtype, lg.dtype)
    if c.dtype == f.dtype and y.dtype == gdt:
        oc, of = K.augment_concat(c, f, y, corr.to(gdt), lg.to(gdt), kbt)
    else:
        oc = K.concat_sites(c, y)
        # (the kernel's coordinate half is given the forces and the gradient: its first output is dropped)
        _, of = K.augment_concat(f, f, lg.to(gdt), corr.to(gdt), lg.to(gdt), kbt)
    return K.like_input(oc, coords), K.like_input(of, coords)


def _need_slice(index) -> None:
    if not isinstance(index, slice):
        raise ValueError("Only slices are allowed for indexing.")


class ForcesTrajectory:
    """Forces without positions."""

    def __init__(self, *, forces) -> None:
        if len(forces.shape) != 3:
            raise ValueError("forces must have 3 dimensions.")
        self.forces = forces

    @property
    def n_sites(self) -> int:
        return self.forces.shape[1]

    @property
    def n_dim(self) -> int:
        return self.forces.shape[2]

    def __len__(self) -> int:
        return len(self.forces)

    def __getitem__(self, index: slice) -> "ForcesTrajectory":
        _need_slice(index)
        return self.__class__(forces=self.forces[index])

    def copy(self) -> "ForcesTrajectory":
        return self.__class__(forces=_copy(self.forces))

    def astype(self, *args, **kwargs) -> "ForcesTrajectory":
        return self.__class__(forces=_astype(self.forces, *args, **kwargs))


class CoordsTrajectory:
    """Positions without forces."""

    def __init__(self, *, coords) -> None:
        if len(coords.shape) != 3:
            raise ValueError("coords must have 3 dimensions.")
        self.coords = coords

    @property
    def n_sites(self) -> int:
        return self.coords.shape[1]

    @property
    def n_dim(self) -> int:
        return self.coords.shape[2]

    def __len__(self) -> int:
        return len(self.coords)

    def __getitem__(self, index: slice) -> "CoordsTrajectory":
        _need_slice(index)
        return self.__class__(coords=self.coords[index])

    def copy(self) -> "CoordsTrajectory":
        return self.__class__(coords=_copy(self.coords))

    def astype(self, *args, **kwargs) -> "CoordsTrajectory":
        return self.__class__(coords=_astype(self.coords, *args, **kwargs))


class Trajectory(CoordsTrajectory, ForcesTrajectory):
    """Coordinates and forces of the same shape (n_frames, n_sites, n_dims)."""

    def __init__(self, *, coords, forces) -> None:
        if tuple(coords.shape) != tuple(forces.shape) or len(coords.shape) != 3:
            raise ValueError("coords and forces must be of same shape.")
        CoordsTrajectory.__init__(self, coords=coords)
        ForcesTrajectory.__init__(self, forces=forces)

    def __getitem__(self, index: slice) -> "Trajectory":
        _need_slice(index)
        return Trajectory(coords=self.coords[index], forces=self.forces[index])

    def copy(self) -> "Trajectory":
        return Trajectory(coords=_copy(self.coords), forces=_copy(self.forces))

    def astype(self, *args, **kwargs) -> "Trajectory":
        return self.__class__(
            coords=_astype(self.coords, *args, **kwargs), forces=_astype(self.forces, *args, **kwargs)
        )


class AugmentedTrajectory(Trajectory):
    r"""Trajectory over (x, y) where y ~ g(.|x) is generated by an Augmenter.

    Forces of the extended ensemble (trajectory/core.py:382-390):
    ``F_y = kbt * grad_y log g``, ``F_x = F + kbt * grad_x log g``; real sites come first,
    generated sites are appended along the site axis.
    """

    def __init__(
        self,
        *,
        coords,
        forces,
        augmenter: Augmenter,
        kbt: float,
        override_first_augment: Optional[Tuple[Any, Any]] = None,
    ) -> None:
        self.augmenter = augmenter
        self.kbt = kbt
        self._real_forces = forces
        self._real_n_sites = coords.shape[1]
        if override_first_augment is None:
            ext_coords, ext_forces = self._augment(coords, forces)
        else:
            ext_coords, ext_forces = override_first_augment
        super().__init__(coords=ext_coords, forces=ext_forces)

    def _augment(self, coords, forces) -> Tuple[Any, Any]:
        fused = getattr(self.augmenter, "augment_trajectory", None)
        if fused is not None:
            return fused(coords, forces, self.kbt)
        aug_coords = self.augmenter.sample(coords)
        real_corr, aug_lgrad = self.augmenter.log_gradient(coords, aug_coords)
        return _augment_concat(coords, forces, aug_coords, real_corr, aug_lgrad, self.kbt)

    @property
    def real_coords(self):
        return self.coords[:, : self._real_n_sites, :]

    @real_coords.setter
    def real_coords(self, value: Any) -> NoReturn:  # noqa: ARG002
        raise ValueError("real_positions cannot be reassigned.")

    @property
    def real_forces(self):
        """Forces of the real sites before any augmentation correction."""
        return self._real_forces

    @real_forces.setter
    def real_forces(self, value: Any) -> NoReturn:  # noqa: ARG002
        raise ValueError("real_forces cannot be reassigned.")

    @property
    def n_real_sites(self) -> int:
        return self._real_n_sites

    @property
    def n_aug_sites(self) -> int:
        return self.coords.shape[1] - self._real_n_sites

    @property
    def real_slice(self) -> slice:
        return slice(0, self.n_real_sites)

    @property
    def aug_slice(self) -> slice:
        return slice(self.n_real_sites, self.n_real_sites + self.n_aug_sites)

    def refresh(self) -> None:
        """Draw new augmenting sites (and forces) in place."""
        self.coords, self.forces = self._augment(coords=self.real_coords, forces=self.real_forces)

    def __getitem__(self, index: slice) -> "AugmentedTrajectory":
        _need_slice(index)
        return AugmentedTrajectory(
            coords=self.real_coords[index],
            forces=self.real_forces[index],
            augmenter=self.augmenter,
            kbt=self.kbt,
            override_first_augment=(self.coords[index], self.forces[index]),
        )

    def copy(self) -> "AugmentedTrajectory":
        return self.__class__(
            coords=_copy(self.real_coords),
            forces=_copy(self.real_forces),
            augmenter=deepcopy(self.augmenter),
            kbt=self.kbt,
            override_first_augment=(_copy(self.coords), _copy(self.forces)),
        )

    def astype(self, *args, **kwargs) -> "AugmentedTrajectory":
        return self.__class__(
            coords=_astype(self.real_coords, *args, **kwargs),
            forces=_astype(self.real_forces, *args, **kwargs),
            augmenter=self.augmenter.astype(*args, **kwargs),
            kbt=self.kbt,
            override_first_augment=(
                _astype(self.coords, *args, **kwargs),
                _astype(self.forces, *args, **kwargs),
            ),
        )

    def pullback(self, C: Callable[["AugmentedTrajectory"], A], array: bool = False) -> Callable:
        """Turn a callable on AugmentedTrajectory into one on Trajectory (or on two arrays)."""
        if array:

            def on_arrays(coords, forces) -> A:
                return C(self.__class__(coords=coords, forces=forces, augmenter=self.augmenter, kbt=self.kbt))

            return on_arrays

        def on_traj(t: Trajectory) -> A:
            return C(self.__class__(coords=t.coords, forces=t.forces, augmenter=self.augmenter, kbt=self.kbt))

        return on_traj

    @classmethod
    def from_trajectory(cls, t: Trajectory, kbt: float, augmenter: Augmenter) -> "AugmentedTrajectory":
        return cls(coords=t.coords, forces=t.forces, augmenter=augmenter, kbt=kbt)
