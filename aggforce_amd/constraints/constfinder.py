"""Guess constrained bonds from coordinate fluctuations (reference: constraints/constfinder.py:14-57).

Pairs of sites whose distance has a standard deviation over the trajectory below ``threshold`` are
taken to be constrained.  The reference materialises all (n_steps, n_sites, n_sites) distances; here
the per-pair variance comes from one streaming GPU pass (K6, ``aggf_pair_dist_var``), so the default
``project_forces(constrained_inds="auto")`` also works on full-size trajectories.  With ``cross_xyz``
(two different systems; not on the force-map path) the small host computation is kept.
"""
from typing import Union

import numpy as np

from ..util import distances
from .hints import Constraints


def guess_pairwise_constraints(xyz, cross_xyz: Union[None, np.ndarray] = None, threshold: float = 1e-3) -> Constraints:
    """Pairs of sites whose distance fluctuates by less than ``threshold`` (standard deviation).

    Returns a set of frozensets {i, j}; with ``cross_xyz`` a set of ordered tuples (i, j) with i
    indexing ``cross_xyz`` and j indexing ``xyz`` (as the reference).
    """
    if cross_xyz is not None:
        x = xyz.detach().cpu().numpy() if hasattr(xyz, "detach") else np.asarray(xyz)
        c = cross_xyz.detach().cpu().numpy() if hasattr(cross_xyz, "detach") else np.asarray(cross_xyz)
        spread = np.std(distances(x, cross_xyz=c), axis=0)
        first, second = np.nonzero(spread < threshold)
        return {(int(i), int(j)) for i, j in zip(first, second)}
    import torch
    from .. import _kernels as K

    var = K.pair_dist_var(K.as_device(xyz))
    close = torch.sqrt(var) < threshold
    close.fill_diagonal_(False)
    idx = torch.nonzero(torch.triu(close, diagonal=1)).cpu().numpy()
    return {frozenset((int(i), int(j))) for i, j in idx}
