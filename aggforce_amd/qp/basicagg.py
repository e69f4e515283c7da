"""Uniform, constraint-aware aggregation map (reference: qp/basicagg.py:11-62).

Pure set logic on an (n_cg, n_fg) matrix -- no trajectory data is touched -- so it stays on
the host.
"""
from typing import Union

import numpy as np

from ..constraints import Constraints, reduce_constraint_sets
from ..map import LinearMap, SeperableTMap
from ..trajectory import ForcesTrajectory


def constraint_aware_uni_map(
    traj: ForcesTrajectory,  # noqa: ARG001  (ignored, as in the reference)
    coord_map: LinearMap,
    constraints: Union[None, Constraints] = None,
) -> SeperableTMap:
    """Sum, with unit weights, the forces of every fg site a cg site depends on, plus those of
    all sites constrained to them."""
    groups = reduce_constraint_sets(set() if constraints is None else constraints)
    matrix = np.zeros_like(coord_map.standard_matrix)
    for cg, row in enumerate(coord_map.standard_matrix):
        members = set(np.nonzero(row)[0].tolist())
        # one sweep over the disjoint groups, each tested against the growing member set,
        # exactly like the reference's product(cg_sets, constraints) loop
        for g in groups:
            if members & g:
                members |= g
        matrix[cg, sorted(members)] = 1.0
    return SeperableTMap(coord_map=coord_map, force_map=LinearMap(matrix))
