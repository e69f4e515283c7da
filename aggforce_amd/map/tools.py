"""Helpers that build special-purpose LinearMaps (reference: map/tools.py)."""
from itertools import combinations
from typing import Iterable, Union

import numpy as np

from ..trajectory.core import AugmentedTrajectory
from .core import LinearMap


def lmap_augvariables(aug: AugmentedTrajectory) -> LinearMap:
    """Slice map selecting the sites an Augmenter added (reference map/tools.py:13-33)."""
    return LinearMap([[site] for site in range(aug.n_real_sites, aug.n_sites)], n_fg_sites=aug.n_sites)


def smear_map(
    site_groups: Iterable[Iterable[int]], n_sites: int, return_mapping_matrix: bool = False
) -> Union[LinearMap, np.ndarray]:
    """(n_sites, n_sites) float32 map replacing every group of sites by the group mean.

    Reference map/tools.py:63-104.  Groups must be disjoint.
    """
    groups = [sorted(set(g)) for g in site_groups]
    for g, h in combinations(groups, 2):
        if set(g) & set(h):
            raise ValueError("Site definitions in site_groups overlap; merge before passing.")
    matrix = np.eye(n_sites, dtype=np.float32)
    for g in groups:
        matrix[np.ix_(g, g)] = 1 / len(g)
    return matrix if return_mapping_matrix else LinearMap(mapping=matrix)
