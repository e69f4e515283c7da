"""aggforce_amd: the aggforce force-map optimisation hot path on AMD MI355X (gfx950).

Drop-in for the hot path of noegroup/aggforce: ``project_forces(coords, forces, coord_map,
constrained_inds, method=...)``, ``LinearMap``, ``Trajectory`` and the ``method=`` plug-ins
``qp_linear_map``, ``qp_feat_linear_map``, ``joptgauss_map``, ``constraint_aware_uni_map``
keep the reference's signatures and result keys; the arithmetic runs in hand-written HIP
kernels (libaggf.so, C ABI in include/aggf.h) on torch ROCm tensors.  There is no CPU
fallback: without the built library and a GPU the compute entry points raise.
"""
from .trajectory import Trajectory
from .agg import project_forces
from .constraints import guess_pairwise_constraints
from .qp import (
    qp_linear_map,
    constraint_aware_uni_map,
    joptgauss_map,
    stagedjoptgauss_map,
    stagedjslicegauss_map,
    stagedjforcegauss_map,
)
from .map import LinearMap

__version__ = "0.1.0"

__all__ = [
    "Trajectory",
    "project_forces",
    "guess_pairwise_constraints",
    "qp_linear_map",
    "constraint_aware_uni_map",
    "joptgauss_map",
    "stagedjoptgauss_map",
    "stagedjslicegauss_map",
    "stagedjforcegauss_map",
    "LinearMap",
]
