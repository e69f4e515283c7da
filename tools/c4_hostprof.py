"""Host-side profile (cProfile) of one c4 (featurised) project_forces step: where the wall time goes when the GPU
kernels only account for half of it.  Run on the GPU box: python tools/c4_hostprof.py"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from aggforce_amd import LinearMap, project_forces
from aggforce_amd import _kernels as K
from aggforce_amd.qp import Multifeaturize, gb_feat, id_feat, qp_feat_linear_map
from aggforce_amd.util import Curry

T, N, n_cg = 20000, 1024, 64
forces = K.synth_normal(T, N, torch.float32, 42100, sigma=30.0)
coords = K.synth_normal(T, N, torch.float32, 42101, sigma=0.3, lattice=1.5)
cons = {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
cmap = LinearMap([[3 * (i * (N // n_cg) // 3)] for i in range(n_cg)], n_fg_sites=N)
kw = dict(method=qp_feat_linear_map, kbt=0.6955215, l2_regularization=10.0, n_constraint_frames=20,
          featurizer=Multifeaturize([id_feat, Curry(gb_feat, outer=8.0, inner=0.0, n_basis=8, width=1.0)]))


def step():
    out = project_forces(coords=coords, forces=forces, coord_map=cmap, constrained_inds=cons,
                         rng=np.random.default_rng(1), **kw)
    torch.cuda.synchronize()
    return out


step()
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
