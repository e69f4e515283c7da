"""Optimal linear force maps (reference: qp/qplinear.py).

For every coarse-grained site i the reference solves, with OSQP,
    x_i = argmin 1/2 x'(C'F'FC + l2 C'C) x   s.t.  (M C) x = e_i,      W_i = C x_i
(qplinear.py:66-88).  All sites share P and A, so here the Gram matrix is built once on
the GPU (K1, MFMA SYRK fused with the constraint-group sums), optionally summed over
frame-sharded ranks with one all-reduce, and all n_cg problems are solved exactly by one
on-device factorisation (K2).
"""
from typing import Union

import numpy as np
from typing_extensions import TypedDict

from .. import _kernels as K
from ..constraints import Constraints, group_layout, groups_csr
from ..distributed import all_reduce_sum_sym_
from ..map import LinearMap, SeperableTMap
from ..trajectory import ForcesTrajectory

SolverOptions = TypedDict(
    "SolverOptions",
    {"solver": str, "eps_abs": float, "max_iter": int, "polish": bool, "polish_refine_iter": int},
    total=False,
)
# Kept for signature compatibility (qplinear.py:21-27).  The on-device solve is exact, so the
# OSQP options are accepted and ignored.
DEFAULT_SOLVER_OPTIONS: SolverOptions = {
    "solver": "osqp",
    "eps_abs": 1e-7,
    "max_iter": int(1e3),
    "polish": True,
    "polish_refine_iter": 10,
}


def qp_form(target):
    """(n_steps, n_sites, 3) -> (n_steps*3, n_sites), rows ordered (step, dim) (qplinear.py:91-103).

    Host/array helper kept for API compatibility; the GPU path never materialises this copy.
    """
    if hasattr(target, "transpose") and hasattr(target, "detach"):
        mixed = target.transpose(1, 2)
        return mixed.reshape(mixed.shape[0] * mixed.shape[1], -1)
    mixed = np.swapaxes(target, 1, 2)
    return np.reshape(mixed, (mixed.shape[0] * mixed.shape[1], -1))


def make_bond_constraint_matrix(n_sites: int, constraints: Constraints) -> np.ndarray:
    """Dense (n_sites, n_reduced) 0/1 matrix expanding reduced coefficients (qplinear.py:106-164)."""
    goa, n_red = group_layout(n_sites, constraints)
    mat = np.zeros((n_sites, n_red))
    mat[np.arange(n_sites), goa] = 1
    return mat


def _one_hot_rows(A: np.ndarray):
    """Column of the single 1 of every row if all rows of A are unit vectors at distinct columns (and A has fewer rows
    than columns), else None."""
    m, n = A.shape
    if m == 0 or m >= n:
        return None
    nz = A != 0
    if not (np.all(nz.sum(axis=1) == 1) and np.all(A[nz] == 1.0)):
        return None
    cols = np.argmax(nz, axis=1).astype(np.int32)
    return cols if len(np.unique(cols)) == m else None


def solve_constrained_maps(G, l2_regularization: float, l2_diag, A_host, what: str = "Map optimization", pins=None,
                           pins_dev=None):
    """Run K2 for all rows of A at once; returns (X device (m, n), stats host).  ``pins`` (int32 array): the rows of
    A are unit vectors at these columns (A_host may then be None); otherwise A_host is inspected for that.
    ``pins_dev``: ``pins`` already on the device of G (a pageable upload here is a blocking copy: the host would wait
    for the Gram kernel in flight before it can queue the solve)."""
    import torch

    dev = G.device
    if pins is None:
        A_host = np.ascontiguousarray(A_host, dtype=np.float64)
        pins = _one_hot_rows(A_host)
        pins_dev = None
    A = None

    def solve(l2, n_refine=1):
        nonlocal A
        if pins is not None:
            # every row of A is a unit vector (slice coordinate map): the constraints pin variables, one
            # factorisation of the free block does it (aggf_eq_qp_solve_pinned)
            nonlocal pins_dev
            if pins_dev is None:
                pins_dev = torch.from_numpy(pins).to(dev)
            return K.eq_qp_solve_pinned(G, l2, l2_diag, pins_dev)
        if A is None:
            A = torch.from_numpy(A_host).to(dev)
        return K.eq_qp_solve(G, l2, l2_diag, A, n_refine=n_refine)

    X, stats = solve(l2_regularization)
    st = stats.cpu().numpy()
    if st[0] > 0 and st[0] <= G.shape[0] and np.isfinite(st[3]):
        # P = G + l2 C'C is singular on null(A): fewer independent frames than free variables and no
        # regularisation.  The minimiser is then not unique, but every minimiser maps the training frames
        # identically (the objective is strictly convex in the mapped forces), and the reference's OSQP hands
        # back one of them.  Do the same: a relative Tikhonov shift of 1e-10 picks the (nearly) minimum-norm one.
        import warnings

        warnings.warn(
            f"{what}: the normal matrix is singular on the feasible set (pivot {int(st[0])}); the force map is not "
            "unique -- returning a minimum-norm minimiser. Add frames or use l2_regularization > 0.",
            stacklevel=3,
        )
        X, stats = solve(l2_regularization + 1e-10 * float(st[3]), n_refine=2)
        st = stats.cpu().numpy()
    if st[0] != 0 or not np.isfinite(st[1]):
        raise ValueError(
            f"{what} failed: the shifted normal matrix is not positive definite "
            f"(pivot {int(st[0])}, constraint residual {st[1]:.3e}). "
            "The problem is under-determined; add frames or use l2_regularization > 0."
        )
    if st[1] > 1e-8:
        # e.g. two CG sites whose atoms were merged into one constraint group: (M C) x = e_i has no
        # solution (the reference's solver returns None there and the map assembly fails)
        raise ValueError(
            f"{what} failed: the equality constraints cannot be met (residual {st[1]:.3e}); "
            "the coordinate map is rank deficient once constrained atoms share a coefficient."
        )
    return X, st


def _upload_together(arrays, device):
    """Device copies of a dict of small host arrays through ONE host-to-device transfer (16-byte aligned pieces of one
    byte buffer, viewed with their own dtypes)."""
    import torch

    offs, total = {}, 0
    for k, a in arrays.items():
        offs[k] = total
        total += -(-a.nbytes // 16) * 16
    buf = np.zeros(max(total, 16), dtype=np.uint8)
    for k, a in arrays.items():
        buf[offs[k]:offs[k] + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    dbuf = torch.from_numpy(buf).to(device)
    tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32,
           np.dtype(np.int64): torch.int64}
    return {k: dbuf[offs[k]:offs[k] + a.nbytes].view(tdt[a.dtype]).reshape(a.shape) for k, a in arrays.items()}


class LinearProblem:
    """Reduced-variable layout of one (coord_map, constraints) pair: Gram, solve and map assembly.

    Shared by ``qp_linear_map`` and the Gram-reusing cross-validation (``agg.project_forces_grid_cv``):
    ``gram`` is K1 on the atoms merged into constraint groups, ``solve`` is K2 for all CG sites,
    ``tmap`` expands the reduced coefficients (n_cg, n_red) to the force map (n_cg, n_fg).
    """

    def __init__(self, coord_map: LinearMap, constraints: Union[None, Constraints], device) -> None:
        import torch

        self.coord_map = coord_map
        self.device = device
        self.n_fg = coord_map.n_fg_sites
        self.goa, self.n_red = group_layout(self.n_fg, constraints if constraints is not None else set())
        self.grp_ptr = self.grp_atoms = None
        self._csr = None
        self._A = None
        host = {"sizes": np.bincount(self.goa, minlength=self.n_red).astype(np.float64), "goa": self.goa}
        if self.n_red != self.n_fg:
            self._csr = groups_csr(self.goa, self.n_red)
            host["grp_ptr"], host["grp_atoms"] = self._csr
        # a slice coordinate map (the map object caches its row -> atom index): A = M C has unit rows at the reduced
        # variables of the mapped atoms -- the pinned variables of aggf_eq_qp_solve_pinned; A itself is not formed
        self.pins = self._pins_d = None
        idx = coord_map._onehot_index() if hasattr(coord_map, "_onehot_index") else None
        if idx is not None and len(idx) < self.n_red:
            pins = self.goa[idx].astype(np.int32)
            if len(np.unique(pins)) == len(pins):
                self.pins = host["pins"] = pins
        # ONE upload for all the index arrays, now, before the Gram kernel is queued: the copy of a pageable array
        # blocks the host until the device is idle (behind the Gram kernel it would hold back the launches of the
        # solve), and five separate copies are five round trips of ~30 us with the device idle
        dev_arrays = _upload_together(host, device)
        self.sizes = dev_arrays["sizes"]
        self._goa_d = dev_arrays["goa"]  # for tmap()
        self.grp_ptr, self.grp_atoms = dev_arrays.get("grp_ptr"), dev_arrays.get("grp_atoms")
        self._pins_d = dev_arrays.get("pins")

    @property
    def A(self) -> np.ndarray:
        """The constraint rows M @ con_mat (qplinear.py:82), (n_cg, n_red): column sums of the coordinate matrix
        over each constraint group, without forming con_mat."""
        if self._A is None:
            M = np.asarray(self.coord_map.standard_matrix, dtype=np.float64)
            self._A = M if self._csr is None else np.add.reduceat(M[:, self._csr[1]], self._csr[0][:-1], axis=1)
        return self._A

    def _goa_dev(self):
        import torch

        if self._goa_d is None:
            self._goa_d = torch.from_numpy(self.goa).to(self.device)
        return self._goa_d

    def gram(self, forces, gram_dtype=None, out=None, accumulate: bool = False):
        """K1 on one block of frames; ``out``/``accumulate`` add to an existing Gram (frame chunks)."""
        import torch

        if forces.shape[1] != self.n_fg:
            raise ValueError(f"forces have {forces.shape[1]} sites but coord_map expects {self.n_fg}")
        cdt = forces.dtype if gram_dtype is None else K.torch_dtype(gram_dtype)
        if forces.dtype == torch.float64:
            cdt = torch.float64
        return K.gram(forces, self.grp_ptr, self.grp_atoms, self.n_red, cdt, out=out, accumulate=accumulate)

    def solve(self, G, l2_regularization: float = 0.0):
        X, _ = solve_constrained_maps(G, float(l2_regularization), self.sizes, self.A if self.pins is None else None,
                                      pins=self.pins, pins_dev=self._pins_d)
        return X

    def tmap(self, X) -> SeperableTMap:
        import torch

        W = K.expand_map(X, self._goa_dev(), self.n_fg)
        return SeperableTMap(coord_map=self.coord_map, force_map=LinearMap.from_device(W))


def qp_linear_map(
    traj: ForcesTrajectory,
    coord_map: LinearMap,
    constraints: Union[None, Constraints] = None,
    l2_regularization: float = 0.0,
    solver_args: SolverOptions = DEFAULT_SOLVER_OPTIONS,  # noqa: ARG001  (exact solve: unused)
    *,
    gram_dtype=None,
    comm=None,
) -> SeperableTMap:
    """Force map minimising the mean squared mapped force (reference qplinear.py:30-88).

    Arguments as in the reference: ``traj`` supplies the forces (n_frames, n_fg, 3),
    ``coord_map`` the configurational map, ``constraints`` a set of frozensets of
    constrained fg indices (atoms of a merged group share one coefficient),
    ``l2_regularization`` penalises the norm of the expanded map.  Extras:
    ``gram_dtype`` (np.float32/np.float64; default: the dtype of the forces) is the
    arithmetic type of the Gram products -- float64 reproduces the reference exactly for
    float32 forces, float32 is the fast MFMA path; ``comm`` is a torch.distributed process
    group (or True for the default group) over which frames are sharded.

    Returns ``SeperableTMap(coord_map, LinearMap(W))`` with ``W`` float64 (n_cg, n_fg).
    """
    forces = K.as_device(traj.forces)
    prob = LinearProblem(coord_map, constraints, forces.device)
    G = prob.gram(forces, gram_dtype)
    all_reduce_sum_sym_(G, comm)
    return prob.tmap(prob.solve(G, l2_regularization))
