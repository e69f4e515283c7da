"""CPU oracle for the aggforce force-map optimisation hot path.

THIS FILE IS TEST INFRASTRUCTURE.  It is a NumPy restatement of what the
reference (noegroup/aggforce, /root/reference) computes on the hot path and is
used ONLY as the checker in ``tests/``, in ``__graft_entry__.smoke()`` and as the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``aggforce_amd/`` imports it;
the product path fails loudly when the HIP library is missing.

Parity status
-------------
* Linear path (Gram, constraint reduction, apply, residual, NaN policy,
  LinearMap algebra, constraint_aware_uni_map, id_feat, qp_feat_linear_map
  plumbing, AugmentedTrajectory algebra, SimpleCondNormal): PINNED against the
  reference itself, imported in the build container by ``oracle/gen_golden.py``
  (fixtures under ``tests/golden/``), and against the reference's own data
  files (water dimer known answer ``tests/test_agg.py:43``, CLN025 force
  matrices ``tests/data/cln_*_force_mat.txt``).
* The QP solve itself lives in a third-party dependency that is absent here
  (``qpsolvers`` -> OSQP, unpinned: ``setup.cfg:25-29``).  ``eq_qp_solve`` restates
  the mathematical problem OSQP is asked to solve at ``qplinear.py:83-85`` /
  ``featlinearmap.py:375-381`` (equality-constrained convex QP) and solves it
  exactly; it is anchored on the reference's known-answer test for that call
  site (water dimer, atol 5e-3).
* ``gb_feat`` values and ``JCondNormal`` (JAX) cannot be imported here:
  PARITY UNPINNED for those two; they are restated from source text
  (``qp/jaxfeat.py``, ``trajectory/jaxgausstraj.py``) and self-checked by finite
  differences and by the identity-premap case of ``SimpleCondNormal``.

Every function cites the reference file:line it follows (paths relative to
``/root/reference/src/aggforce``).
"""
from __future__ import annotations

import copy
from itertools import product
from typing import Dict, FrozenSet, Iterable, List, Optional, Sequence, Set, Tuple

import numpy as np

Constraints = Set[FrozenSet[int]]

# ----------------------------------------------------------------------------
# constraints (constraints/tools.py)
# ----------------------------------------------------------------------------


def reduce_constraint_sets(constraints: Constraints) -> Constraints:
    """Merge overlapping constraint sets into disjoint ones.

    Follows constraints/tools.py:7-77 including its control flow (pop, flood,
    "second try"), because the *iteration order* of the returned set feeds the
    label order of ``id_feat``.
    """
    constraints_copy = copy.copy(constraints)
    agged: Constraints = set()
    if len(constraints) <= 1:
        return constraints_copy
    new = frozenset(constraints_copy.pop())
    second_try = False
    while True:
        to_add = [x for x in constraints_copy if new.intersection(x)]
        new = new.union(*to_add)
        constraints_copy.difference_update(to_add)
        if not to_add:
            agged.add(new)
            if second_try:
                second_try = False
                try:
                    new = frozenset(constraints_copy.pop())
                except KeyError:
                    break
            else:
                second_try = True
    return agged


def constraint_lookup_dict(constraints: Constraints) -> Dict[int, int]:
    """member -> anchor (smallest member) map; constraints/tools.py:80-116."""
    mapping: Dict[int, int] = {}
    for group in constraints:
        sites = sorted(group)
        for s in sites[1:]:
            mapping[s] = sites[0]
    return mapping


def make_bond_constraint_matrix(n_sites: int, constraints: Constraints) -> np.ndarray:
    """C in {0,1}^(N x n_red); qp/qplinear.py:106-164."""
    rcons = reduce_constraint_sets(constraints)
    n_constrained = sum(len(x) for x in rcons)
    n_red = n_sites - n_constrained + len(rcons)
    lookup = constraint_lookup_dict(rcons)
    mat = np.zeros((n_sites, n_red))
    offset = 0
    for site in range(n_sites):
        if site not in lookup:
            mat[site, offset] = 1
            offset += 1
    for site, anchor in lookup.items():
        mat[site, :] = mat[anchor, :]
    return mat


def distances(xyz: np.ndarray, cross_xyz: Optional[np.ndarray] = None) -> np.ndarray:
    """Per-frame distance matrices; util.py:65-72 (matrix form only)."""
    if cross_xyz is None:
        disp = xyz[:, None, :, :] - xyz[:, :, None, :]
    else:
        disp = xyz[:, None, :, :] - cross_xyz[:, :, None, :]
    return np.linalg.norm(disp, axis=-1)


def guess_pairwise_constraints(xyz: np.ndarray, threshold: float = 1e-3) -> Constraints:
    """Pairs whose distance std-dev is below threshold; constraints/constfinder.py:46-53."""
    dists = distances(xyz)
    sds = np.sqrt(np.var(dists, axis=0))
    np.fill_diagonal(sds, threshold * 2)
    inds = np.nonzero(sds < threshold)
    return {frozenset(int(i) for i in v) for v in zip(*inds)}


# ----------------------------------------------------------------------------
# array primitives (util.py, map/core.py, agg.py)
# ----------------------------------------------------------------------------


def qp_form(target: np.ndarray) -> np.ndarray:
    """(T,N,3) -> (3T,N), row=(t,d); qp/qplinear.py:91-103."""
    mixed = np.swapaxes(target, 1, 2)
    return np.reshape(mixed, (mixed.shape[0] * mixed.shape[1], -1))


def trjdot(points: np.ndarray, factor: np.ndarray) -> np.ndarray:
    """einsum('tfd,cf->tcd') or per-frame factor; util.py:119-125."""
    # same einsum call (incl. the explicit contraction path) as the reference, so that the
    # CPU baseline in bench.py times what the reference would execute
    opt_path = ["einsum_path", (0, 1)]
    if factor.ndim == 2:
        return np.einsum("tfd,cf->tcd", points, factor, optimize=opt_path)
    if factor.ndim == 3:
        return np.einsum("...fd,...cf->...cd", points, factor, optimize=opt_path)
    raise ValueError("Factor matrix is an incompatible shape.")


def has_nans(x: np.ndarray) -> bool:
    """map/core.py:13-16."""
    flat = x.ravel(order="K")
    return bool(np.isnan(np.dot(flat, flat)))


def linearmap_apply(
    points: np.ndarray,
    matrix: np.ndarray,
    handle_nans=True,
    nan_check_threshold: float = 1e-6,
) -> np.ndarray:
    """LinearMap.__call__ incl. NaN policy; map/core.py:219-240.

    Never mutates ``points`` (the reference's temporary in-place edit is undone
    before it returns, so the observable result is the same).
    """
    if handle_nans and has_nans(points):
        mask = np.isnan(points)
        work = points.copy()
        work[mask] = 0.0
        raw = trjdot(work, matrix)
        work[mask] = -1.0
        pushed = trjdot(work, matrix)
        if not np.allclose(raw, pushed, atol=nan_check_threshold):
            raise ValueError(
                "NaN handling is on and results seem to depend on NaN "
                "positions in input array. Check input and standard_matrix."
            )
        return raw
    return trjdot(points, matrix)


def list_mapping_matrix(mapping: Sequence[Sequence[int]], n_fg_sites: int) -> np.ndarray:
    """list-of-lists LinearMap constructor; map/core.py:133-144."""
    mat = np.zeros((len(mapping), n_fg_sites))
    for site, contents in enumerate(mapping):
        local = np.zeros(n_fg_sites)
        local[list(contents)] = 1 / len(contents)
        mat[site, :] = local
    return mat


def force_smoothness(array: np.ndarray) -> float:
    """mean squared element; agg.py:291-297."""
    return float(np.mean(array**2))


# ----------------------------------------------------------------------------
# exact equality-constrained QP (what qpsolvers/OSQP is asked to solve)
# ----------------------------------------------------------------------------


def eq_qp_solve(
    P: np.ndarray, q: Optional[np.ndarray], A: np.ndarray, b: np.ndarray
) -> np.ndarray:
    """argmin 1/2 x'Px + q'x  s.t. Ax=b, exactly, by the null-space method.

    Problem statement of the call sites qp/qplinear.py:83-85 and
    qp/featlinearmap.py:375-381 (qpsolvers.solve_qp(P,q,A=A,b=b)).  ``b`` may be a
    vector or a matrix of right-hand sides (one column per problem).  Rank
    deficient / redundant ``A`` rows are handled through the SVD (consistent
    systems assumed; the least-squares particular solution is used otherwise).
    A singular reduced Hessian gives the minimum-norm reduced solution.
    """
    P = np.asarray(P, dtype=np.float64)
    A = np.asarray(A, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    n = P.shape[0]
    vec = b.ndim == 1
    B = b[:, None] if vec else b
    qv = np.zeros((n, 1)) if q is None else np.asarray(q, dtype=np.float64).reshape(n, 1)
    # row-scale invariant SVD of A
    U, s, Vt = np.linalg.svd(A, full_matrices=True)
    tol = max(A.shape) * np.finfo(np.float64).eps * (s[0] if s.size else 0.0)
    r = int(np.sum(s > tol))
    V1 = Vt[:r].T  # range(A')
    Z = Vt[r:].T  # null(A)
    xp = V1 @ ((U[:, :r].T @ B) / s[:r, None])
    if Z.shape[1] == 0:
        X = xp
    else:
        H = Z.T @ P @ Z
        H = 0.5 * (H + H.T)
        g = Z.T @ (P @ xp + qv)
        # scale-aware solve; Cholesky when PD, min-norm lstsq otherwise
        try:
            L = np.linalg.cholesky(H)
            y = np.linalg.solve(L, -g)
            z = np.linalg.solve(L.T, y)
            # one refinement step
            res = -g - H @ z
            z = z + np.linalg.solve(L.T, np.linalg.solve(L, res))
        except np.linalg.LinAlgError:
            z = np.linalg.lstsq(H, -g, rcond=None)[0]
        X = xp + Z @ z
    return X[:, 0] if vec else X


# ----------------------------------------------------------------------------
# linear force-map optimisation (qp/qplinear.py)
# ----------------------------------------------------------------------------


def linear_problem(
    forces: np.ndarray,
    coord_matrix: np.ndarray,
    constraints: Optional[Constraints],
    l2_regularization: float = 0.0,
) -> Dict[str, np.ndarray]:
    """All intermediates of qp_linear_map up to the solver call; qplinear.py:63-82."""
    if constraints is None:
        constraints = set()
    reshaped = qp_form(forces)
    con_mat = make_bond_constraint_matrix(coord_matrix.shape[1], constraints)
    reg_mat = np.matmul(reshaped, con_mat)
    qp_mat = np.matmul(reg_mat.T, reg_mat)
    if l2_regularization > 0.0:
        qp_mat = qp_mat + l2_regularization * np.matmul(con_mat.T, con_mat)
    constraint_mat = np.matmul(coord_matrix, con_mat)
    return {"con_mat": con_mat, "reg_mat": reg_mat, "qp_mat": qp_mat, "A": constraint_mat}


def qp_linear_map(
    forces: np.ndarray,
    coord_matrix: np.ndarray,
    constraints: Optional[Constraints] = None,
    l2_regularization: float = 0.0,
) -> np.ndarray:
    """Optimal force-map matrix W (n_cg x N); qplinear.py:30-88.

    The n_cg problems share P and A, so they are solved as one multi-RHS
    problem (mathematically identical to the reference's per-site loop).
    """
    pr = linear_problem(forces, coord_matrix, constraints, l2_regularization)
    n_cg = coord_matrix.shape[0]
    X = eq_qp_solve(pr["qp_mat"], None, pr["A"], np.eye(n_cg))
    return (pr["con_mat"] @ X).T


def project_forces(
    coords: np.ndarray,
    forces: np.ndarray,
    coord_matrix: np.ndarray,
    constraints: Optional[Constraints] = None,
    l2_regularization: float = 0.0,
    handle_nans=True,
) -> Dict[str, object]:
    """agg.py:49-136 with method=qp_linear_map (constraints given explicitly)."""
    W = qp_linear_map(forces, coord_matrix, constraints, l2_regularization)
    mapped_coords = linearmap_apply(coords, coord_matrix, handle_nans=handle_nans)
    mapped_forces = linearmap_apply(forces, W, handle_nans=True)
    return {
        "mapped_coords": mapped_coords,
        "mapped_forces": mapped_forces,
        "force_map": W,
        "residual": force_smoothness(mapped_forces),
    }


def constraint_aware_uni_map(
    coord_matrix: np.ndarray, constraints: Optional[Constraints]
) -> np.ndarray:
    """Uniform constraint-aware aggregation matrix; qp/basicagg.py:44-61."""
    if constraints is None:
        constraints = set()
    cg_sets = [set(np.nonzero(row)[0]) for row in coord_matrix]
    constraints = reduce_constraint_sets(constraints)
    for group, x in product(cg_sets, constraints):
        if group.intersection(x):
            group.update(x)
    out = np.zeros_like(coord_matrix)
    for ind, contents in enumerate(cg_sets):
        out[ind, list(contents)] = 1.0
    return out


# ----------------------------------------------------------------------------
# featurised path (qp/featlinearmap.py, qp/jaxfeat.py)
# ----------------------------------------------------------------------------


def id_feat_ids(n_fg_sites: int, constraints: Constraints) -> np.ndarray:
    """Constraint-group label of every atom; featlinearmap.py:598-609.

    ``sorted`` on disjoint frozensets keeps CPython's set iteration order, so
    this must be (and is) the same Python expression as the reference's.
    """
    groups = copy.deepcopy(constraints)
    groups = groups.union(frozenset([x]) for x in range(n_fg_sites))
    reduced_groups = sorted(reduce_constraint_sets(groups))
    ids = np.zeros(n_fg_sites, dtype=np.int32)
    for label, fg_set in enumerate(reduced_groups):
        ids[list(fg_set)] = label
    return ids


def id_feat(n_frames: int, n_fg_sites: int, constraints: Constraints):
    """One-hot group features (T,N,G) float32 and zero divs; featlinearmap.py:598-627."""
    ids = id_feat_ids(n_fg_sites, constraints)
    n_types = int(ids.max()) + 1 if n_fg_sites else 0
    feats = np.zeros((n_frames, n_fg_sites, n_types), dtype=np.float32)
    feats[:, np.arange(n_fg_sites), ids] = 1
    divs = np.zeros((n_frames, n_types, 3), dtype=np.float32)
    return feats, divs


def smear_matrix(site_groups: Iterable[Iterable[int]], n_sites: int) -> np.ndarray:
    """Group-mean projector (N,N) float32; map/tools.py:94-101."""
    matrix = np.zeros((n_sites, n_sites), dtype=np.float32)
    np.fill_diagonal(matrix, 1)
    for group in [set(x) for x in site_groups]:
        inds0, inds1 = zip(*product(group, group))
        matrix[inds0, inds1] = 1 / len(group)
    return matrix


def gb_centers(outer: float, inner: float, n_basis: int, dist_power: float, dtype=np.float32):
    """Gaussian grid centres; jaxfeat.py:235-236."""
    pow_grid = np.linspace(inner**dist_power, outer**dist_power, n_basis).astype(dtype)
    return (pow_grid ** dtype(1 / dist_power)).astype(dtype)


def gb_feat_site(
    points: np.ndarray,
    cg_points_site: np.ndarray,
    ids: np.ndarray,
    smear: np.ndarray,
    outer: float,
    inner: float = 0.0,
    n_basis: int = 10,
    width: float = 1.0,
    dist_power: float = 0.5,
    clip: float = 1e-3,
    n_channels: Optional[int] = None,
    dtype=np.float32,
):
    """gb_subfeat + gb_subfeat_jac('reorder') for ONE cg site; jaxfeat.py:371-565.

    points (T,N,3); cg_points_site (T,3).  Returns feats (T,N,n_basis*n_channels)
    and divs (T,n_basis*n_channels,3) in ``dtype`` (JAX default float32).
    ``n_channels`` defaults to the reference's ``max(ids)`` (jaxfeat.py:115 -- one
    less than the number of labels: the last label's slice falls out of range
    and JAX drops the update, so that channel is all zero / absent).
    The divergence is the closed form of jacrev(sum of Gaussians) summed over the
    atoms of each channel, with cg_points held constant (closure constant at
    jaxfeat.py:546-559).
    """
    points = np.asarray(points, dtype=dtype)
    cg = np.asarray(cg_points_site, dtype=dtype)
    T, N, _ = points.shape
    if n_channels is None:
        n_channels = int(ids.max())
    centers = gb_centers(outer, inner, n_basis, dist_power, dtype)
    p = np.einsum("tfd,cf->tcd", points, smear.astype(dtype))  # jaxfeat.py:449-450
    disp = p - cg[:, None, :]
    r = np.linalg.norm(disp, axis=-1)  # (T,N)
    arg = (r[..., None] - centers[None, None, :]) / dtype(width)
    raw = np.exp(-(arg**2))
    gauss = np.maximum(raw, dtype(clip)) - dtype(clip)  # clipped_gauss 272-276
    # d gauss / d r  (zero where clipped)
    dgauss = np.where(raw > clip, -2.0 * arg / dtype(width) * raw, 0.0).astype(dtype)
    n_feat = n_basis * n_channels
    feats = np.zeros((T, N, n_feat), dtype=dtype)
    divs = np.zeros((T, n_feat, 3), dtype=dtype)
    with np.errstate(invalid="ignore", divide="ignore"):
        unit = disp / r[..., None]  # NaN at r == 0, as jnp.linalg.norm's gradient
    # d/dx_a sum_{a'} g_k(r_{a'}) = sum_{a'} g'_k(r_{a'}) unit_{a'} S[a',a]
    per_atom_grad = np.einsum("tbk,tbd,ba->takd", dgauss, unit, smear.astype(dtype))
    for a in range(N):
        ch = int(ids[a])
        if ch >= n_channels:
            continue  # dropped channel (quirk A)
        sl = slice(n_basis * ch, n_basis * (ch + 1))
        feats[:, a, sl] = gauss[:, a, :]
        divs[:, sl, :] += per_atom_grad[:, a, :, :]
    return feats, divs


def feat_constraint_arrays(
    feat: np.ndarray, cg_ind: int, coord_matrix: np.ndarray, frame_indices: np.ndarray
):
    """_constr_arrays with the sampled frames injected; featlinearmap.py:445-459."""
    sub = feat[frame_indices]
    mult = np.einsum("ca,...af->...cf", coord_matrix, sub)
    target = np.zeros((len(frame_indices), coord_matrix.shape[0]))
    target[:, cg_ind] = 1
    return mult.reshape((-1, mult.shape[-1])), target.reshape((-1,))


def feat_site_problem(forces, feat, div, kbt, l2_regularization):
    """reg_mat and qp_mat of one cg site; featlinearmap.py:361-372."""
    force_features = np.einsum("...af,...ad->...fd", forces, feat)
    ms_reg_mat = force_features + kbt * np.swapaxes(div, 1, 2)
    reg_mat = np.reshape(ms_reg_mat, (-1, ms_reg_mat.shape[2]))
    qp_mat = np.matmul(reg_mat.T, reg_mat)
    if l2_regularization > 0:
        qp_mat = qp_mat + np.diag((l2_regularization,) * qp_mat.shape[0])
    return reg_mat, qp_mat


def qp_feat_linear_map(
    forces: np.ndarray,
    coord_matrix: np.ndarray,
    feats: Sequence[np.ndarray],
    divs: Sequence[np.ndarray],
    kbt: float,
    frame_indices: Sequence[np.ndarray],
    l2_regularization: float = 1e1,
) -> List[np.ndarray]:
    """Per-site feature coefficients; featlinearmap.py:349-384."""
    coefs = []
    for ind, (feat, div) in enumerate(zip(feats, divs)):
        A, b = feat_constraint_arrays(feat, ind, coord_matrix, np.asarray(frame_indices[ind]))
        _, qp_mat = feat_site_problem(forces, feat, div, kbt, l2_regularization)
        coefs.append(eq_qp_solve(qp_mat, None, A, b))
    return coefs


def cla_apply(forces, feats, divs, coefs) -> np.ndarray:
    """CLAMap.__call__ with scale_f/trans_f; featlinearmap.py:512-520, map/core.py:428-430."""
    scale = np.stack([np.einsum("...ij,j->...i", f, c) for f, c in zip(feats, coefs)], axis=1)
    trans = np.stack([np.einsum("tij,i->tj", d, c) for d, c in zip(divs, coefs)], axis=1)
    return trjdot(forces, scale) + trans


# ----------------------------------------------------------------------------
# noised ("Gaussian") maps (trajectory/*, qp/jgauss.py)
# ----------------------------------------------------------------------------


def condnormal_log_gradient(source, generated, premap_matrix, var, dtype=np.float32):
    """Closed form of JCondNormal.log_gradient with cov = var*I; jaxgausstraj.py:263-284.

    y ~ N(Mx, var I): d/dy log g = -(y - Mx)/var ; d/dx log g = M'(y - Mx)/var.
    Reduces to SimpleCondNormal.log_gradient (simplegausstraj.py:108-110) when M = I.
    """
    source = np.asarray(source, dtype=dtype)
    generated = np.asarray(generated, dtype=dtype)
    M = np.asarray(premap_matrix, dtype=dtype)
    resid = (generated - trjdot(source, M)) / dtype(var)
    d_gen = -resid
    d_src = np.einsum("tcd,cf->tfd", resid, M)
    return d_src.astype(dtype), d_gen.astype(dtype)


def condnormal_sample(source, premap_matrix, var, noise, dtype=np.float32):
    """y = Mx + sqrt(var) * eps with injected standard-normal eps; jaxgausstraj.py:232-234,311-319."""
    source = np.asarray(source, dtype=dtype)
    M = np.asarray(premap_matrix, dtype=dtype)
    return (trjdot(source, M) + dtype(np.sqrt(var)) * np.asarray(noise, dtype=dtype)).astype(dtype)


def condnormal_full_sample(source, premap_matrix, cov, noise, dtype=np.float64):
    """JCondNormal.sample with a FULL covariance over the flattened generated coordinates (site-major, xyz innermost):
    multivariate_normal(mean = premap(flat source), cov), jaxgausstraj.py:232-234,311-316, with the variate written as
    mean + L eps, L L' = cov (Cholesky: JAX's default ``method``), eps injected.  PARITY UNPINNED (JAX absent)."""
    source = np.asarray(source, dtype=dtype)
    mean = trjdot(source, np.asarray(premap_matrix, dtype=dtype))
    L = np.linalg.cholesky(np.asarray(cov, dtype=np.float64)).astype(dtype)
    flat = mean.reshape(len(mean), -1) + np.asarray(noise, dtype=dtype).reshape(len(mean), -1) @ L.T
    return flat.reshape(mean.shape).astype(dtype)


def condnormal_full_log_gradient(source, generated, premap_matrix, cov, dtype=np.float64):
    """grad of logpdf(y; premap(x), cov) with respect to (x, y): jaxgausstraj.py:77-96 in closed form,
    d/dy = -cov^-1 (y - Mx), d/dx = (M (x) I_3)' cov^-1 (y - Mx).  PARITY UNPINNED (JAX absent)."""
    source = np.asarray(source, dtype=dtype)
    generated = np.asarray(generated, dtype=dtype)
    M = np.asarray(premap_matrix, dtype=dtype)
    resid = (generated - trjdot(source, M)).reshape(len(source), -1)
    r = (resid @ np.linalg.inv(np.asarray(cov, dtype=np.float64)).astype(dtype)).reshape(generated.shape)
    return np.einsum("tcd,cf->tfd", r, M).astype(dtype), (-r).astype(dtype)


def augment(coords, forces, premap_matrix, var, kbt, noise, dtype=np.float32):
    """AugmentedTrajectory._augment; trajectory/core.py:382-390."""
    aug_coords = condnormal_sample(coords, premap_matrix, var, noise, dtype)
    real_corr, aug_lgrad = condnormal_log_gradient(coords, aug_coords, premap_matrix, var, dtype)
    aug_forces = kbt * aug_lgrad
    real_forces = forces + kbt * real_corr
    full_coords = np.concatenate([coords, aug_coords], axis=1)
    full_forces = np.concatenate([real_forces, aug_forces], axis=1)
    return full_coords, full_forces


def joptgauss_force_map(coords, forces, coord_matrix, var, kbt, noise, constraints=None,
                        l2_regularization=0.0, dtype=np.float32):
    """joptgauss_map up to the optimised augmented force map; qp/jgauss.py:114-131."""
    full_coords, full_forces = augment(coords, forces, coord_matrix, var, kbt, noise, dtype)
    n_real = coords.shape[1]
    n_aug = coord_matrix.shape[0]
    aug_cmap = list_mapping_matrix([[x] for x in range(n_real, n_real + n_aug)], n_real + n_aug)
    W = qp_linear_map(full_forces, aug_cmap, constraints, l2_regularization)
    return {"aug_coords": full_coords, "aug_forces": full_forces, "aug_coord_matrix": aug_cmap,
            "force_map": W}


# ---- staged Gaussian maps (qp/jgauss.py:143-650) -- PARITY UNPINNED (JAX absent here) ---------
def _augment_postmap(coords, forces, var, kbt, noise, postmap, dtype=np.float32):
    """AugmentedTrajectory._augment with JCondNormal(cov=var, source_postmap=Q): identity premap,
    source log-gradient passed through Q (jaxgausstraj.py:281-283)."""
    n = coords.shape[1]
    eye = np.eye(n, dtype=dtype)
    aug_coords = condnormal_sample(coords, eye, var, noise, dtype)
    d_src, d_gen = condnormal_log_gradient(coords, aug_coords, eye, var, dtype)
    if postmap is not None:
        d_src = trjdot(d_src, np.asarray(postmap, dtype=dtype))
    full_coords = np.concatenate([np.asarray(coords, dtype=dtype), aug_coords], axis=1)
    full_forces = np.concatenate([forces + kbt * d_src, kbt * d_gen], axis=1)
    return full_coords, full_forces


def staged_gauss_fit(coords, forces, coord_matrix, var, kbt, noise, variant="opt", force_matrix=None,
                     constraints=None, l2_regularization=0.0, premap_l2_regularization=0.0,
                     dtype=np.float32):
    """Fit of stagedjoptgauss_map ("opt", jgauss.py:216-263), stagedjforcegauss_map ("force",
    jgauss.py:531-599) or stagedjslicegauss_map ("slice", jgauss.py:378-427).

    Returns the pieces of the ComposedTMap: pre coordinate/force matrices, the second-stage
    force matrix over [mapped real | noise] sites, and the source_postmap Q = W M'.
    """
    M = np.asarray(coord_matrix, dtype=np.float64)
    n_cg = M.shape[0]
    if variant == "slice":
        W = np.ones_like(M)
        f_in = np.full_like(np.asarray(coords, dtype=dtype), np.nan)
    else:
        W = (np.asarray(force_matrix, dtype=np.float64) if force_matrix is not None
             else qp_linear_map(forces, M, constraints, premap_l2_regularization))
        f_in = np.zeros_like(forces) if variant == "force" else forces
    full_coords, full_forces = augment(coords, f_in, M.astype(dtype), var, kbt, noise, dtype)
    n_real = coords.shape[1]
    # RATMap: real sites through the pre-map (no NaN handling for the slice variant's ones map)
    pm_coords = np.concatenate([trjdot(full_coords[:, :n_real], M), full_coords[:, n_real:]], axis=1)
    pm_forces = np.concatenate([trjdot(full_forces[:, :n_real], W), full_forces[:, n_real:]], axis=1)
    slice_map = list_mapping_matrix([[x] for x in range(n_cg, 2 * n_cg)], 2 * n_cg)
    if variant == "slice":
        W2 = constraint_aware_uni_map(slice_map, set())
        Q = None
    else:
        W2 = qp_linear_map(pm_forces, slice_map, set(), l2_regularization)
        Q = W @ M.T
    return {"pre_coord_matrix": M, "pre_force_matrix": W, "post_force_matrix": W2, "postmap": Q,
            "pmapped_coords": pm_coords, "pmapped_forces": pm_forces}


def staged_gauss_apply(fit, coords, forces, var, kbt, noise, variant="opt", dtype=np.float32):
    """Application of the ComposedTMap returned by the staged maps to (coords, forces)."""
    M, W, W2 = fit["pre_coord_matrix"], fit["pre_force_matrix"], fit["post_force_matrix"]
    n_cg = M.shape[0]
    if variant == "slice":
        forces = np.full_like(np.asarray(coords, dtype=np.float64), np.nan)
    c1 = trjdot(coords, M)
    f1 = trjdot(forces, W)
    full_coords, full_forces = _augment_postmap(c1, f1, var, kbt, noise, fit["postmap"], dtype)
    slice_map = list_mapping_matrix([[x] for x in range(n_cg, 2 * n_cg)], 2 * n_cg)
    return trjdot(full_coords, slice_map), linearmap_apply(full_forces, W2)


def project_forces_grid_cv(grid_l2, coords, forces, coord_matrix, folds, constraints=None):
    """Intent of project_forces_grid_cv for the linear optimiser; agg.py:185-234.

    ``folds``: list of frame-index arrays (the reference shuffles with an unseeded generator, so
    the split is an input here).  Returns {l2: (mean score, sample sd, n_runs)} with the hold-out
    score force_smoothness(W F_val) (agg.py:224-227; ``from_arrays`` there is read as ``map_arrays``).
    """
    out = {}
    for l2 in grid_l2:
        scores = []
        for k, val in enumerate(folds):
            train = np.concatenate([f for j, f in enumerate(folds) if j != k])
            W = qp_linear_map(forces[train], coord_matrix, constraints, l2)
            scores.append(float(np.mean(linearmap_apply(forces[val], W) ** 2)))
        m = sum(scores) / len(scores)
        sd = (sum((s - m) ** 2 for s in scores) / (len(scores) - 1)) ** 0.5
        out[l2] = (m, sd, len(scores))
    return out
