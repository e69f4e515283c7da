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


def guess_pairwise_constraints(xyz, cross_xyz: Union[None, np.ndarray] = None, threshold: float = 1e-3,
                               comm=None) -> Constraints:
    """Pairs of sites whose distance fluctuates by less than ``threshold`` (standard deviation).

    Returns a set of frozensets {i, j}; with ``cross_xyz`` a set of ordered tuples (i, j) with i
    indexing ``cross_xyz`` and j indexing ``xyz`` (as the reference).  ``comm`` (extra): ``xyz`` is this
    rank's shard of a frame-sharded trajectory; the per-rank means and variances are combined exactly
    (two all-reduces of (N, N)), so every rank gets the set the whole trajectory gives.
    """
    if cross_xyz is not None:
        x = xyz.detach().cpu().numpy() if hasattr(xyz, "detach") else np.asarray(xyz)
        c = cross_xyz.detach().cpu().numpy() if hasattr(cross_xyz, "detach") else np.asarray(cross_xyz)
        spread = np.std(distances(x, cross_xyz=c), axis=0)
        first, second = np.nonzero(spread < threshold)
        return {(int(i), int(j)) for i, j in zip(first, second)}
    import torch
    from .. import _kernels as K

    from ..distributed import all_reduce_sum_, world_size

    x = K.as_device(xyz)
    if world_size(comm) > 1:
        # exact pooling of the per-rank (n_r, mean_r, var_r):  var = sum_r (n_r / n) (var_r + (mean_r - mean)^2)
        mean_r, var_r = K.pair_dist_moments(x)
        n = torch.full((1,), float(x.shape[0]), dtype=torch.float64, device=x.device)
        all_reduce_sum_(n, comm)
        weight = float(x.shape[0]) / float(n.item())
        mean = K.axpby(weight, mean_r, 0.0, mean_r)
        all_reduce_sum_(mean, comm)
        var = K.pair_pool_term(var_r, mean_r, mean, weight)
        all_reduce_sum_(var, comm)
    else:
        var = K.pair_dist_var(x)
    close = var < float(threshold) * float(threshold)  # std < threshold
    close.fill_diagonal_(False)
    idx = torch.nonzero(torch.triu(close, diagonal=1)).cpu().numpy()
    return {frozenset((int(i), int(j))) for i, j in idx}

