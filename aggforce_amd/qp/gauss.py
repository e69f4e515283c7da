"""Optimised Gaussian ("noised") maps (reference: qp/jgauss.py:27-140).

``joptgauss_map`` extends the trajectory with Gaussian-noised copies of the mapped sites
(K5, fused augmentation kernel), optimises a linear force map on the extended system with
``qp_linear_map`` (K1 + K2 on N + n_cg sites) and wraps the result so that it acts on plain
trajectories.  The JAX pieces of the reference (JLinearMap, JCondNormal autodiff) are
replaced by the closed-form augmenter ``CondNormal``.
"""
import warnings
from typing import Optional

import numpy as np

from .. import _kernels as K
from ..constraints import Constraints
from ..map import (
    AugmentedTMap,
    ComposedTMap,
    LinearMap,
    NullForcesTMap,
    RATMap,
    SeperableTMap,
    lmap_augvariables,
)
from ..trajectory import AugmentedTrajectory, CondNormal, CoordsTrajectory, Trajectory
from .basicagg import constraint_aware_uni_map
from .qplinear import DEFAULT_SOLVER_OPTIONS, SolverOptions, qp_linear_map


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
    fused = _joptgauss_without_extended_arrays(traj, augmenter, kbt, constraints, kwargs)
    if fused is not None:
        return fused
    aug_traj = AugmentedTrajectory.from_trajectory(t=traj, augmenter=augmenter, kbt=kbt)
    # constraints are index sets over the real sites, which keep their indices (the generated
    # sites are appended at the end)
    aug_tmap = qp_linear_map(
        traj=aug_traj, coord_map=lmap_augvariables(aug_traj), constraints=constraints, **kwargs
    )
    return AugmentedTMap(aug_tmap=aug_tmap, augmenter=augmenter, kbt=kbt)


def _joptgauss_without_extended_arrays(traj, augmenter, kbt, constraints, kwargs):
    """joptgauss_map's fit without the (n_frames, N + n_cg, 3) arrays, or None when the layout does not allow it.

    The extended forces are [F - Fa C | Fa] (Fa = the generated sites' forces, C = the premap), so their Gram
    matrix is Tm' Gx Tm with Gx the Gram matrix of [F | Fa] -- K1 reads F where it lies and the small Fa beside it
    (``aggf_gram_pair``), ``aggf_augmented_gram`` applies Tm, constraint groups are summed on the matrix
    (``aggf_sym_group_reduce``).  Round 2 wrote and re-read 2 x 13 GB of copies per fit at BASELINE config 4's size
    (and again per application): 19.5 % of the step.  Needs N and n_cg to be multiples of 128 (K1's in-place
    layout), float32/float64 forces in the augmenter's promoted dtype and products in that dtype; anything else
    takes the general path."""
    from ..distributed import all_reduce_sum_sym_
    from .qplinear import LinearProblem

    extra = set(kwargs) - {"l2_regularization", "solver_args", "gram_dtype", "comm"}
    if extra or not isinstance(traj, Trajectory) or isinstance(traj, AugmentedTrajectory):
        return None
    import torch

    from ..distributed import agree_on_min

    comm = kwargs.get("comm")
    forces = K.as_device(traj.forces)
    coords = K.as_device(traj.coords)
    n_real, n_aug = forces.shape[1], augmenter.premap_map(forces.shape[1]).n_cg_sites
    gram_dtype = kwargs.get("gram_dtype")
    # A shard without frames contributes a zero matrix (only under ``comm``: alone it is the general path's error).
    empty = forces.shape[0] == 0
    ok = not ((empty and comm is None) or n_real % 128 or n_aug % 128
              or (gram_dtype is not None and K.torch_dtype(gram_dtype) != forces.dtype)
              or torch.promote_types(coords.dtype, K.torch_dtype(augmenter.dtype)) != forces.dtype
              or (not empty and (not forces.is_contiguous() or forces.data_ptr() % 16)))
    # The choice rests on rank-local facts (contiguity, alignment, dtypes, an empty shard), and the two paths
    # all-reduce DIFFERENT matrices (Gx of [F | Fa] here, the Gram of [F - Fa C | Fa] in qp_linear_map): every rank
    # must take the same one, as cv_joptgauss_fold_grams does.
    if not agree_on_min(int(ok), comm, forces.device):
        return None
    # the reduced-variable layout first: its index arrays go up in one blocking copy, which behind the Gram kernel would
    # wait for it and leave the host ~0.5 ms of set-up (the slice map over N + n_cg sites, the layout) to do with the
    # device idle before the solve can be queued
    aug_cmap = LinearMap(mapping=[[i] for i in range(n_real, n_real + n_aug)], n_fg_sites=n_real + n_aug)
    prob = LinearProblem(aug_cmap, constraints, forces.device)
    if empty:
        cols = augmenter.correction_columns(n_real, forces.device)
        Gx = torch.zeros((n_real + n_aug, n_real + n_aug), dtype=torch.float64, device=forces.device)
    else:
        y, fa, cols = augmenter.noise_sites(coords, kbt)
        assert K.gram_pair_ok(forces, fa)
        del y
        Gx = K.gram_pair(forces, fa)
    all_reduce_sum_sym_(Gx, comm)  # linear in Gx: the transform commutes with the sum over ranks
    G = K.augmented_gram(Gx, n_real, cols)
    del Gx
    if prob.grp_ptr is not None:
        G = K.sym_group_reduce(G, prob.grp_ptr, prob.grp_atoms, prob.n_red)
    aug_tmap = prob.tmap(prob.solve(G, float(kwargs.get("l2_regularization", 0.0))))
    return AugmentedTMap(aug_tmap=aug_tmap, augmenter=augmenter, kbt=kbt)


def cv_joptgauss_fold_grams(coords, forces, coord_map: LinearMap, var: float, kbt: float, constraints, seed, folds,
                            gram_dtype=None, noise=None, comm=None):
    """Per-fold Gram matrices of joptgauss_map's extended system for ``project_forces_grid_cv``'s one-pass form, or
    None when the layout does not allow the in-place fit (see ``_joptgauss_without_extended_arrays``).

    ONE noise realisation is drawn for every frame (the reference's loop draws afresh for every fit and every
    application: the same distribution, and here every grid point and fold sees common random numbers), fold by
    fold on the gathered frames of the fold, and fold k's matrix is ``Tm' Gram([F | Fa] on fold k) Tm`` with the
    constraint groups summed -- exactly the matrix a fit on those frames alone would form.  ``comm``: the frames are
    sharded over ranks and every rank splits ITS frames into the folds (global fold k = the union of the ranks' fold
    k, as in the linear one-pass form); the matrices are all-reduced, each rank draws its own noise (seed offset by
    the rank).  Returns (fold_grams (k, n, n), LinearProblem of the extended system)."""
    import torch

    from ..distributed import agree_on_min, all_reduce_sum_sym_, rank_of
    from .qplinear import LinearProblem

    f_dev, c_dev = K.as_device(forces), K.as_device(coords)
    n_real, n_aug = f_dev.shape[1], coord_map.n_cg_sites
    if seed is not None and comm is not None:
        seed = int(seed) + 7919 * rank_of(comm)
    augmenter = CondNormal(var=var, premap=coord_map, seed=seed)
    ok = not (f_dev.shape[0] == 0 or n_real % 128 or n_aug % 128
              or (gram_dtype is not None and K.torch_dtype(gram_dtype) != f_dev.dtype)
              or torch.promote_types(c_dev.dtype, K.torch_dtype(augmenter.dtype)) != f_dev.dtype
              or not f_dev.is_contiguous() or any(len(idx) == 0 for idx in folds))
    if not agree_on_min(int(ok), comm, f_dev.device):  # every rank takes the same path
        return None
    aug_cmap = LinearMap(mapping=[[i] for i in range(n_real, n_real + n_aug)], n_fg_sites=n_real + n_aug)
    prob = LinearProblem(aug_cmap, constraints, f_dev.device)
    grams = []
    for idx in folds:  # one fold's frames at a time: the gathered copies never exceed a fold
        fk, ck = K.take_frames(f_dev, idx), K.take_frames(c_dev, idx)
        if noise is not None:
            augmenter.inject_noise(np.asarray(noise)[np.asarray(idx)])
        y, fak, cols = augmenter.noise_sites(ck, kbt)  # (every call advances the augmenter's stream: independent draws)
        del y, ck
        assert K.gram_pair_ok(fk, fak)  # (implied by the layout test above)
        G = K.augmented_gram(K.gram_pair(fk, fak), n_real, cols)
        if prob.grp_ptr is not None:
            G = K.sym_group_reduce(G, prob.grp_ptr, prob.grp_atoms, prob.n_red)
        grams.append(G)
        del fk, fak
    out = torch.stack(grams)
    all_reduce_sum_sym_(out, comm)
    return out, prob


# ---- staged maps: deterministic linear pre-map, then a noising step --------------------------
def _noise_site_slice_map(n_sites: int, n_aug: int) -> LinearMap:
    """Coordinate map isolating the last ``n_aug`` sites of a partially mapped trajectory."""
    return LinearMap(mapping=[[i] for i in range(n_sites - n_aug, n_sites)], n_fg_sites=n_sites)


def _pre_tmap(traj, coord_map, force_map, constraints, premap_l2_regularization, premap_solver_args):
    if force_map is None:
        return qp_linear_map(
            traj=traj,
            coord_map=coord_map,
            constraints=constraints,
            l2_regularization=premap_l2_regularization,
            solver_args=premap_solver_args,
        )
    return SeperableTMap(coord_map=coord_map, force_map=force_map)


def _backmapped_noise_postmap(pre_tmap) -> LinearMap:
    """W M' : carries the noise force on the mapped real sites through the force map.

    grad_x f(Mx) = M' [grad f](Mx) is the atomistic noise force; the force map W then maps it
    (reference jgauss.py:266-280, ``j_force_map @ j_coord_map.T``).
    """
    W = pre_tmap.force_map.standard_matrix
    M = pre_tmap.coord_map.standard_matrix
    return LinearMap(W @ M.T, handle_nans=False)


def stagedjoptgauss_map(
    traj: Trajectory,
    coord_map: LinearMap,
    var: float,
    kbt: float,
    force_map: Optional[LinearMap] = None,
    constraints: Optional[Constraints] = None,
    seed: Optional[int] = None,
    premap_l2_regularization: float = 0.0,
    premap_solver_args: SolverOptions = DEFAULT_SOLVER_OPTIONS,
    noise=None,
    **kwargs,
) -> ComposedTMap:
    """Optimised Gaussian map as a linear pre-map followed by a noising map (jgauss.py:143-312).

    Steps: (1) optimise (or take ``force_map`` as) a noise-free force map; (2) augment the
    full-resolution trajectory with noised copies of the mapped sites (K5); (3) map its real
    sites with the pre-map (K3), keeping the noise sites; (4) optimise a second force map that
    mixes mapped-real and noise forces (K1 + K2 on 2 n_cg sites, no constraints).  The result
    is ``ComposedTMap([post_tmap, pre_tmap])``: ``pre_tmap`` may be applied before saving data,
    ``post_tmap`` (stochastic) afterwards.  ``noise`` as in ``joptgauss_map``: the first array
    is consumed by the fit, the following ones by successive applications of ``post_tmap``.
    """
    pre_tmap = _pre_tmap(traj, coord_map, force_map, constraints, premap_l2_regularization,
                         premap_solver_args)
    augmenter = CondNormal(var=var, premap=pre_tmap.coord_map, seed=seed)
    noise = list(noise) if noise is not None else []
    if noise:
        augmenter.inject_noise(noise[0])
    aug_traj = AugmentedTrajectory.from_trajectory(t=traj, augmenter=augmenter, kbt=kbt)
    pmapped_traj = RATMap(tmap=pre_tmap)(aug_traj)
    pmapped_tmap = qp_linear_map(
        traj=pmapped_traj,
        coord_map=_noise_site_slice_map(pmapped_traj.n_sites, aug_traj.n_aug_sites),
        constraints=set(),
        **kwargs,
    )
    pmapped_augmenter = CondNormal(var=var, source_postmap=_backmapped_noise_postmap(pre_tmap), seed=seed)
    if len(noise) > 1:
        pmapped_augmenter.inject_noise(*noise[1:])
    post_tmap = AugmentedTMap(aug_tmap=pmapped_tmap, augmenter=pmapped_augmenter, kbt=kbt)
    return ComposedTMap(submaps=[post_tmap, pre_tmap])


def stagedjslicegauss_map(
    traj: CoordsTrajectory,
    coord_map: LinearMap,
    var: float,
    kbt: float,
    seed: Optional[int] = None,
    constraints: Optional[Constraints] = None,  # noqa: ARG001
    warn_input_forces: bool = True,
    noise=None,
) -> ComposedTMap:
    """Gaussian map whose reported forces come from the noise only (jgauss.py:315-446).

    Three submaps: ``[2]`` adds null (NaN) forces so that force-free input is accepted, ``[1]``
    maps the coordinates to the coarse resolution, ``[0]`` noises them and reports the
    noise-site forces.  No force information of ``traj`` is used.
    """
    naforce_traj = NullForcesTMap(warn_input_forces=warn_input_forces)(traj)
    augmenter = CondNormal(var=var, premap=coord_map, seed=seed)
    noise = list(noise) if noise is not None else []
    if noise:
        augmenter.inject_noise(noise[0])
    aug_traj = AugmentedTrajectory.from_trajectory(t=naforce_traj, augmenter=augmenter, kbt=kbt)
    null_fmap = LinearMap(mapping=np.ones_like(coord_map.standard_matrix), handle_nans=False)
    pre_tmap = SeperableTMap(coord_map=coord_map, force_map=null_fmap)
    pmapped_traj = RATMap(tmap=pre_tmap)(aug_traj)
    pmapped_tmap = constraint_aware_uni_map(
        traj=pmapped_traj,
        coord_map=_noise_site_slice_map(pmapped_traj.n_sites, aug_traj.n_aug_sites),
        constraints=set(),
    )
    pmapped_augmenter = CondNormal(var=var, seed=seed)
    if len(noise) > 1:
        pmapped_augmenter.inject_noise(*noise[1:])
    post_tmap = AugmentedTMap(aug_tmap=pmapped_tmap, augmenter=pmapped_augmenter, kbt=kbt)
    return ComposedTMap(submaps=[post_tmap, pre_tmap, NullForcesTMap(warn_input_forces=False)])


def stagedjforcegauss_map(
    traj: Trajectory,
    coord_map: LinearMap,
    var: float,
    kbt: float,
    force_map: Optional[LinearMap] = None,
    constraints: Optional[Constraints] = None,
    seed: Optional[int] = None,
    premap_l2_regularization: float = 0.0,
    premap_solver_args: SolverOptions = DEFAULT_SOLVER_OPTIONS,
    contribution_tolerance: float = 1e-6,
    noise=None,
    **kwargs,
) -> ComposedTMap:
    """Gaussian map with the least possible noise-derived force content (jgauss.py:449-650).

    As ``stagedjoptgauss_map``, but the second optimisation sees a trajectory whose real
    forces are zero, so it minimises the noise contribution alone; a warning is raised when the
    remaining mean-square noise force exceeds ``contribution_tolerance``.
    """
    pre_tmap = _pre_tmap(traj, coord_map, force_map, constraints, premap_l2_regularization,
                         premap_solver_args)
    augmenter = CondNormal(var=var, premap=pre_tmap.coord_map, seed=seed)
    noise = list(noise) if noise is not None else []
    if noise:
        augmenter.inject_noise(noise[0])
    zeroforce_traj = Trajectory(coords=traj.coords, forces=K.scaled(traj.forces, 0))
    aug_traj = AugmentedTrajectory.from_trajectory(t=zeroforce_traj, augmenter=augmenter, kbt=kbt)
    pmapped_traj = RATMap(tmap=pre_tmap)(aug_traj)
    pmapped_tmap = qp_linear_map(
        traj=pmapped_traj,
        coord_map=_noise_site_slice_map(pmapped_traj.n_sites, aug_traj.n_aug_sites),
        constraints=set(),
        **kwargs,
    )
    mapped_forces = pmapped_tmap(pmapped_traj).forces
    remaining_force_residual = float(K.sumsq(K.as_device(mapped_forces))) / max(1, int(np.prod(mapped_forces.shape)))
    if remaining_force_residual > contribution_tolerance:
        warnings.warn(
            "Unable to remove all noise contributions in forces. Remaining "
            f"contribution: {remaining_force_residual}.",
            stacklevel=0,
        )
    pmapped_augmenter = CondNormal(var=var, source_postmap=_backmapped_noise_postmap(pre_tmap), seed=seed)
    if len(noise) > 1:
        pmapped_augmenter.inject_noise(*noise[1:])
    post_tmap = AugmentedTMap(aug_tmap=pmapped_tmap, augmenter=pmapped_augmenter, kbt=kbt)
    return ComposedTMap(submaps=[post_tmap, pre_tmap])
