"""Guess constrained bonds from coordinate fluctuations (host helper, small inputs).

Reference: constraints/constfinder.py:14-57.  The reference's tests and examples call
this with <= 10 frames; it is the step *before* the hot path
and is kept as plain host arithmetic in this round.
"""
from typing import Union

import numpy as np

from ..util import distances
from .hints import Constraints


def guess_pairwise_constraints(
    xyz, cross_xyz: Union[None, np.ndarray] = None, threshold: float = 1e-3
) -> Constraints:
    """Pairs of sites whose distance has a standard deviation below ``threshold``."""
    if hasattr(xyz, "detach"):
        xyz = xyz.detach().cpu().numpy()
    if cross_xyz is not None and hasattr(cross_xyz, "detach"):
        cross_xyz = cross_xyz.detach().cpu().numpy()
    spread = np.std(distances(xyz, cross_xyz=cross_xyz), axis=0)
    if cross_xyz is None:
        np.fill_diagonal(spread, 2 * threshold)
        first, second = np.nonzero(spread < threshold)
        return {frozenset((int(i), int(j))) for i, j in zip(first, second)}
    first, second = np.nonzero(spread < threshold)
    return {(int(i), int(j)) for i, j in zip(first, second)}
