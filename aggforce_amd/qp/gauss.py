"""Optimised Gaussian ("noised") maps (reference: qp/jgauss.py:27-140).

``joptgauss_map`` extends the trajectory with Gaussian-noised copies of the mapped sites
(K5, fused augmentation kernel), optimises a linear force map on the extended system with
``qp_linear_map`` (K1 + K2 on N + n_cg sites) and wraps the result so that it acts on plain
trajectories.  The JAX pieces of the reference (JLinearMap, JCondNormal autodiff) are
replaced by the closed-form augmenter ``CondNormal``.
"""
from typing import Optional

from ..constraints import Constraints
from ..map import AugmentedTMap, LinearMap, lmap_augvariables
from ..trajectory import AugmentedTrajectory, CondNormal, Trajectory
from .qplinear import qp_linear_map


def joptgauss_map(
    traj: Trajectory,
    coord_map: LinearMap,
    var: float,
    kbt: float,
    constraints: Optional[Constraints] = None,
    seed: Optional[int] = None,
    noise=None,
    frame_offset: int = 0,
    **kwargs,
) -> AugmentedTMap:
    """Gaussian map with the optimal mix of real and noise-derived forces.

    Arguments as in the reference: ``var`` is the (scalar, diagonal) noise variance, ``kbt``
    converts log-density gradients into forces, ``seed`` seeds the noise, ``**kwargs`` go to
    ``qp_linear_map``.  Extras: ``noise`` -- a list of standard-normal arrays
    (n_frames, n_cg, 3) consumed by successive augmentations (fit first, then each
    application), for reproducible tests; ``frame_offset`` -- global index of the first
    frame of this rank's shard, so sharded runs draw the same noise as a single-GPU run.

    The returned TMap is stochastic and not separable: use it on Trajectory objects or via
    ``map_arrays``.
    """
    augmenter = CondNormal(var=var, premap=coord_map, seed=seed, frame_offset=frame_offset)
    if noise is not None:
        augmenter.inject_noise(*noise)
    aug_traj = AugmentedTrajectory.from_trajectory(t=traj, augmenter=augmenter, kbt=kbt)
    # constraints are index sets over the real sites, which keep their indices (the generated
    # sites are appended at the end)
    aug_tmap = qp_linear_map(
        traj=aug_traj, coord_map=lmap_augvariables(aug_traj), constraints=constraints, **kwargs
    )
    return AugmentedTMap(aug_tmap=aug_tmap, augmenter=augmenter, kbt=kbt)
