"""Featurised (configuration-dependent) force maps (reference: qp/featlinearmap.py).

The force map of cg site c is linear in user-provided per-atom features:
``W_c(t)[a] = feat_c[t,a,:] . coef_c``.  For every site the reference builds
    R[(t,d), f] = sum_a F[t,a,d] feat[t,a,f] + kbt * div[t,f,d],   P = R'R + l2 I
(featlinearmap.py:361-372) and solves ``min 1/2 x'Px  s.t.  A x = b`` where the rows of A
pin the mapped weights on a sample of frames (``_constr_arrays``, 397-459).

Here R is produced directly in the (t, f, d) layout the Gram kernel K1 consumes, P comes
from the same MFMA SYRK as the linear path, and K2 solves the constrained problem on the
device (with a Tikhonov-regularised, iteratively refined Schur complement, because the
sampled constraint rows are redundant by construction).  Any featuriser following the
reference's protocol works (dense feature arrays, contracted with a batched GEMM); the
built-in ``id_feat``/``gb_feat`` pair has a fused path that never materialises the one-hot
feature tensor (see qp/gbfeat.py).
"""
from typing import Any, Callable, Generator, Iterable, List, Optional, Union

import numpy as np
from typing_extensions import TypedDict

from .. import _kernels as K
from ..constraints import Constraints, reduce_constraint_sets
from ..distributed import agree_on_indices, all_reduce_sum_sym_, shard_extent, take_global_frames
from ..map import CLAFTMap, CLAMap, LinearMap
from ..trajectory import Trajectory
from .qplinear import DEFAULT_SOLVER_OPTIONS, SolverOptions

KNAME_FEATS = "feats"
KNAME_DIVS = "divs"
KNAME_NAMES = "names"

Features = TypedDict(
    "Features",
    {"feats": Iterable[Any], "divs": Iterable[Any], "names": Union[Iterable[str], None]},
)
Featurizer = Callable[[Any, LinearMap, Constraints], Features]
GeneralizedFeatures = Union[Features, "FeatZipper"]
GeneralizedFeaturizer = Callable[[Any, LinearMap, Constraints], GeneralizedFeatures]


def _cat(arrays, axis: int):
    if any(hasattr(a, "detach") for a in arrays):
        import torch

        arrays = [K.as_device(a) for a in arrays]
        dt = arrays[0].dtype
        for a in arrays[1:]:
            dt = torch.promote_types(dt, a.dtype)
        return torch.cat([a.to(dt) for a in arrays], dim=axis)
    return np.concatenate(arrays, axis=axis)


class FeatZipper:
    """Lazily concatenates the per-site output of several featurisers (featlinearmap.py:73-246).

    Indexing with "feats"/"divs" gives a generator over cg sites whose items are the member
    featurisers' arrays joined along the feature axis (axis 2 for feats, axis 1 for divs);
    "names" gives None.  Member iterables are advanced only as items are requested.
    """

    generator_keys = frozenset([KNAME_FEATS, KNAME_DIVS])
    name_key = KNAME_NAMES
    _axis = {KNAME_FEATS: 2, KNAME_DIVS: 1}

    def __init__(self, content: List[GeneralizedFeatures]) -> None:
        self.reset(content)
        self.names = None

    def keys(self) -> frozenset:
        return self.generator_keys | frozenset([KNAME_NAMES])

    def reset(self, content: Iterable[GeneralizedFeatures]) -> None:
        content = list(content)
        self.source = {key: zip(*[member[key] for member in content]) for key in self.generator_keys}

    def _generate(self, key: str) -> Generator[Any, None, None]:
        for pieces in self.source[key]:
            yield _cat(list(pieces), self._axis[key])

    def __getitem__(self, key: str):
        if key in self.generator_keys:
            return self._generate(key)
        if key == KNAME_NAMES:
            return self.names
        raise KeyError("Invalid key; valid keys are {}".format(self.keys()))


def multifeaturize(featurizers: List[GeneralizedFeaturizer]) -> GeneralizedFeaturizer:
    """Closure form of Multifeaturize (featlinearmap.py:630-671)."""

    def composite(copoints, coord_map: LinearMap, constraints: Constraints) -> GeneralizedFeatures:
        return FeatZipper(content=[f(copoints, coord_map, constraints) for f in featurizers])

    return composite


class Multifeaturize:
    """Callable combining featurisers lazily into one (featlinearmap.py:674-745)."""

    def __init__(self, featurizers: Iterable[GeneralizedFeaturizer]) -> None:
        self.featurizers = featurizers

    def __call__(self, *args, **kwargs) -> GeneralizedFeatures:
        return FeatZipper(content=[f(*args, **kwargs) for f in self.featurizers])

    @property
    def fused_fit(self):
        """Fused fitting routine if the members are [id_feat], [gb_feat bound with Curry] or
        [id_feat, gb_feat bound with Curry] (see qp/gbfeat.py); None otherwise."""
        from .gbfeat import fit_id_gb, recognise

        rec = recognise(self.featurizers)
        if rec is None:
            return None
        use_id, gb_kwargs = rec

        def fit(traj, coord_map, kbt, n_constraint_frames, constraints, l2, frame_indices, rng, comm):
            return fit_id_gb(traj, coord_map, kbt, n_constraint_frames, constraints, l2, frame_indices, rng,
                             comm, use_id, gb_kwargs, self)

        return fit

    @property
    def fused_cv(self):
        """One-pass cross-validation over ``l2_regularization`` for the same member lists as ``fused_fit``
        (``qp/gbfeat.py:cv_id_gb``, used by ``project_forces_grid_cv``); None otherwise."""
        from .gbfeat import cv_id_gb, recognise

        rec = recognise(self.featurizers)
        if rec is None:
            return None
        use_id, gb_kwargs = rec

        def cv(coords, forces, coord_map, kbt, n_constraint_frames, constraints, l2_values, folds, rng):
            return cv_id_gb(coords, forces, coord_map, kbt, n_constraint_frames, constraints, l2_values, folds, rng,
                            use_id, gb_kwargs)

        return cv

    def __repr__(self) -> str:
        parts = ["{}():".format(self.__class__)]
        for i, f in enumerate(self.featurizers):
            parts += ["C{}:".format(i), repr(f)]
        return " ".join(parts)

    def __str__(self) -> str:
        lines = ["{} instance:".format(self.__class__)]
        for i, f in enumerate(self.featurizers):
            lines.append("Callable {}:".format(i))
            lines += ["    " + s for s in str(f).split("\n")]
        return "\n".join(lines)


def _reference_group_order(n_fg_sites: int, constraints: Constraints) -> List[frozenset]:
    """Merged constraint groups (singletons included) in the order the reference labels them.

    The reference's labels are positions in ``sorted(reduce_constraint_sets(groups))``
    (featlinearmap.py:598-602).  ``<`` between disjoint frozensets is never true, so that sort
    keeps the iteration order of the set the flood search returns -- and that order is a
    property of CPython's hash tables (frozensets of ints hash deterministically, so it is
    reproducible): which seed ``pop()`` yields next, and where a completed group lands in the
    result set, depend on every insertion, removal and *resize* the containers went through
    (``difference_update`` compacts a table once a quarter of its slots are tombstones, which
    reshuffles what ``pop()`` sees next).  Bit-identical labels therefore need the same trace
    of container operations as constraints/tools.py:49-77 on the same kind of container:

    * the working set: ``constraints`` rebuilt in iteration order (what ``deepcopy`` does to a
      set), then the singletons ``{0}, {1}, ...`` added in site order (featlinearmap.py:598-599);
    * a ``.copy()`` of it is the pool; per merged group: one ``pop()``, then one
      ``difference_update`` per flood round -- the rounds that absorb something and the two
      empty ones that end a group -- and one ``add`` of the finished group to the result set.

    Which member sets a round absorbs is looked up through a site -> member-set index here
    (the reference rescans the whole pool each round, O(N^2) for N sites); only the trace above
    matters for the order, and it is the same.
    """
    working = set(list(constraints)).union(frozenset([site]) for site in range(n_fg_sites))
    pool = working.copy()
    if len(working) <= 1:  # returned untouched by the reference (constraints/tools.py:52-53)
        return sorted(pool)
    sets_of_site: dict = {}
    for member_set in working:
        for site in member_set:
            sets_of_site.setdefault(site, []).append(member_set)
    finished: set = set()
    while pool:
        grown = set(pool.pop())
        frontier = list(grown)
        quiet_rounds = 0
        while quiet_rounds < 2:
            absorbed = {m for site in frontier for m in sets_of_site.get(site, ()) if m in pool}
            pool.difference_update(list(absorbed))
            frontier = [site for m in absorbed for site in m if site not in grown]
            grown.update(frontier)
            if not absorbed:
                quiet_rounds += 1
        finished.add(frozenset(grown))
    return sorted(finished)


def constraint_group_labels(n_fg_sites: int, constraints: Constraints) -> np.ndarray:
    """int32 label of every fg site; constrained sites share a label.

    Bit-identical to the reference's ``id_feat(..., return_ids=True)`` (featlinearmap.py:598-609),
    including its label ORDER (see ``_reference_group_order``): the order decides which feature
    column belongs to which group and, through ``max_channels = max(ids)`` (jaxfeat.py:115),
    which group ``gb_feat`` leaves without Gaussian features.
    """
    ids = np.zeros(n_fg_sites, dtype=np.int32)
    for label, members in enumerate(_reference_group_order(n_fg_sites, constraints)):
        ids[list(members)] = label
    return ids


def id_feat(points, cmap: LinearMap, constraints: Constraints, return_ids: bool = False):
    """One-hot constraint-group label of every fg site as features (featlinearmap.py:553-627).

    Returns ``{"feats": [(T, N, G) float32] * n_cg, "divs": [(T, G, 3) zeros] * n_cg, "names":
    None}`` -- the same array object for every cg site -- or the int32 labels if ``return_ids``.
    """
    ids = constraint_group_labels(cmap.n_fg_sites, constraints)
    if return_ids:
        return ids
    n_frames = points.shape[0]
    n_types = int(ids.max()) + 1 if ids.size else 0
    feats = np.zeros((n_frames, cmap.n_fg_sites, n_types), dtype=np.float32)
    feats[:, np.arange(cmap.n_fg_sites), ids] = 1
    divs = np.zeros((n_frames, n_types, cmap.n_dim), dtype=np.float32)
    return {KNAME_FEATS: [feats] * cmap.n_cg_sites, KNAME_DIVS: [divs] * cmap.n_cg_sites, KNAME_NAMES: None}


# ----------------------------------------------------------------------------------------


def _constraint_rows(feat_dev, cg_ind: int, M_dev, frame_idx, comm=None):
    """A[(s,c), f] = sum_a M[c,a] feat[s,a,f] on the sampled frames, and the one-hot target b
    (reference _constr_arrays, featlinearmap.py:445-459) -- kernel K4b.  ``frame_idx`` numbers
    frames over the whole trajectory; with frames sharded over ``comm`` the sampled feature
    frames are first assembled identically on every rank."""
    idx = np.asarray(frame_idx, dtype=np.int64).reshape(-1)
    if comm is not None:
        sub = take_global_frames(feat_dev, idx, comm)
        return K.feat_constraint_rows(sub, np.arange(idx.size), M_dev, cg_ind)
    return K.feat_constraint_rows(feat_dev, idx, M_dev, cg_ind)


def _site_regression(forces_dev, feat_dev, div_dev, kbt: float, ld: Optional[int] = None):
    """R3[t,f,d] = sum_a feat[t,a,f] F[t,a,d] + kbt div[t,f,d]  -- (T, ld >= n_feat, 3), kernel K4c."""
    return K.feat_contract(forces_dev, feat_dev, div_dev, kbt, ld)


def qp_feat_linear_map(
    traj: Trajectory,
    coord_map: LinearMap,
    featurizer: Featurizer,
    kbt: float,
    n_constraint_frames: int = 20,
    constraints: Union[None, Constraints] = None,
    sparse: bool = True,  # noqa: ARG001  (matrices stay dense on the device)
    solver_args: SolverOptions = DEFAULT_SOLVER_OPTIONS,  # noqa: ARG001  (exact solve)
    l2_regularization: float = 1e1,
    *,
    frame_indices: Optional[List[np.ndarray]] = None,
    rng=None,
    comm=None,
    fused: bool = True,
) -> CLAFTMap:
    """Force map linear in features, minimising the mean squared mapped force
    (reference featlinearmap.py:249-394; same arguments).

    ``featurizer(coords, coord_map, constraints)`` returns {"feats": per-site (T, N, n_feat),
    "divs": per-site (T, n_feat, 3), "names"}.  Extras: ``frame_indices`` (one index array per
    cg site) or ``rng`` (numpy Generator) make the sampled constraint frames reproducible --
    the reference draws them from an unseeded generator (featlinearmap.py:445); ``comm`` shards
    frames over ranks: frame indices then number the WHOLE trajectory, rank 0's draw is used on
    every rank and each sampled frame comes from the rank that owns it, so all ranks solve the
    same problem and hold the same coefficients; ``fused=False`` forces the generic
    dense-feature path even for the built-in featurisers.

    Returns ``CLAFTMap(coord_map, CLAMap)`` with tags {"feat_names", "coef_list"}.
    """
    import torch

    if constraints is None:
        constraints = set()
    fused_fit = getattr(featurizer, "fused_fit", None) if fused else None
    if fused_fit is not None:
        return fused_fit(traj, coord_map, kbt, n_constraint_frames, constraints, l2_regularization,
                         frame_indices, rng, comm)
    feat_results = featurizer(traj.coords, coord_map, constraints)
    feats, divs, names = (feat_results[k] for k in (KNAME_FEATS, KNAME_DIVS, KNAME_NAMES))
    forces = K.as_device(traj.forces)
    dev = forces.device
    M_dev = torch.from_numpy(np.asarray(coord_map.standard_matrix, dtype=np.float64)).to(dev)
    gen = np.random.default_rng() if rng is None else rng
    coefs: List[np.ndarray] = []
    used_frames: List[np.ndarray] = []
    _, n_frames_total = shard_extent(forces.shape[0], comm, dev)
    for ind, (feat, div) in enumerate(zip(feats, divs)):
        feat_dev = K.as_device(feat)
        div_dev = K.as_device(div)
        if frame_indices is not None:
            idx = np.asarray(frame_indices[ind])
        else:
            idx = gen.choice(n_frames_total, size=n_constraint_frames, replace=False)
        idx = agree_on_indices(idx, comm, dev)
        used_frames.append(idx)
        A, b = _constraint_rows(feat_dev, ind, M_dev, idx, comm)
        n_feat = feat_dev.shape[2]
        r3 = _site_regression(forces, feat_dev, div_dev, kbt, ld=-(-n_feat // 128) * 128)  # K1's in-place layout
        G = K.gram(r3, None, None, n_feat, torch.float64)  # exact Gram of R (see qp/gbfeat.py)
        all_reduce_sum_sym_(G, comm)
        X, stats = K.eq_qp_solve(G, float(l2_regularization), None, A, b, schur_reg=1e-12, n_refine=3)
        st = stats.cpu().numpy()
        if st[0] != 0 or not np.isfinite(st[1]):
            raise ValueError(
                f"Map optimization failed. (site {ind}: pivot {int(st[0])}, "
                f"constraint residual {st[1]:.3e}, before refinement {st[2]:.3e}, scale {st[3]:.3e})"
            )
        coefs.append(X[0].cpu().numpy())
        del feat_dev, div_dev, r3, G
    force_map = _feat_linear_mapping(
        featurizer=featurizer,
        coefs=coefs,
        mapping=coord_map,
        constraints=constraints,
        tags={"feat_names": names, "coef_list": coefs, "constraint_frames": used_frames},
    )
    return CLAFTMap(coord_map=coord_map, force_map=force_map)


def _feat_linear_mapping(featurizer, coefs: List[np.ndarray], mapping: LinearMap, constraints, zeroes_check: bool = True,
                         **kwargs) -> CLAMap:
    """CLAMap of a feature-linear map (reference featlinearmap.py:462-530).

    ``scale``/``trans`` follow the reference (each re-runs the featuriser).  ``apply`` is the
    fused form used when the map is called: per site
    ``out[t,c,:] = coef_c . (feat_c[t]' F[t] + div_c[t])`` -- note: no kbt on the divergence
    term here, exactly as in the reference's trans_f (featlinearmap.py:517-520).
    All contractions are HIP kernels (K4c ``aggf_feat_weights`` / ``aggf_feat_contract``, K3).
    """
    import torch

    def _coef_dev(c, device):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(c, dtype=np.float64))).to(device)

    def scale_f(copoints):
        feats = featurizer(copoints, mapping, constraints)[KNAME_FEATS]
        out = None
        for site, (feat, c) in enumerate(zip(feats, coefs)):
            f = K.as_device(feat)
            if out is None:
                out = torch.empty((f.shape[0], len(coefs), f.shape[1]), dtype=torch.float64, device=f.device)
            K.feat_weights(f, _coef_dev(c, f.device), out, site)
        return K.like_input(out, copoints)

    def trans_f(copoints):
        divs = featurizer(copoints, mapping, constraints)[KNAME_DIVS]
        cols = []
        for div, c in zip(divs, coefs):
            d = K.as_device(div, torch.float64)  # (T, n_feat, 3); trans[t,:] = sum_f coef[f] div[t,f,:]
            cols.append(K.linearmap_apply(d, _coef_dev(c, d.device).reshape(1, -1)))
        return K.like_input(torch.cat(cols, dim=1), copoints)

    def apply_f(points, copoints):
        res = featurizer(copoints, mapping, constraints)
        F = K.as_device(points)
        cols = []
        for feat, div, c in zip(res[KNAME_FEATS], res[KNAME_DIVS], coefs):
            y = _site_regression(F, K.as_device(feat), K.as_device(div), 1.0)
            cols.append(K.linearmap_apply(y, _coef_dev(c, F.device).reshape(1, -1)))
        return K.like_input(torch.cat(cols, dim=1), points)

    return CLAMap(scale=scale_f, trans=trans_f, n_fg_sites=mapping.n_fg_sites, zeroes_check=zeroes_check,
                  n_cg_sites=None if zeroes_check else len(coefs), apply=apply_f, **kwargs)
