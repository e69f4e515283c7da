"""Gaussian-basis distance featuriser ``gb_feat`` on the GPU (reference: qp/jaxfeat.py).

Every fine-grained site is characterised by its distance to the mapped (coarse-grained) site,
expanded in clipped Gaussians, one block of ``n_basis`` features per constraint group
("channel"); atoms of a constraint group are first replaced by the group mean, so they share
one feature row (jaxfeat.py:20-184).  The reference builds this with JAX (jit, jacrev); here
the distances, Gaussians and the closed-form divergence come from the HIP kernels K4
(``aggf_gb_channels`` / ``aggf_gb_regmat`` / ``aggf_gb_apply``).

Two ways to use it, both drop-in:

* ``gb_feat(points, cmap, constraints, outer=..., ...)`` returns the reference's featuriser
  dictionary {"feats": per-site (T, N, n_feat), "divs": per-site (T, n_feat, 3), "names": None}
  (dense arrays, float32) -- the generic protocol of ``qp_feat_linear_map``.
* ``Multifeaturize([id_feat, Curry(gb_feat, ...)])`` (or ``[Curry(gb_feat, ...)]``, or
  ``[id_feat]``) is recognised by ``qp_feat_linear_map`` and fitted by the fused path
  (``fit_id_gb``): the regression matrix of each site is written directly in the Gram kernel's
  layout from per-group force sums, never materialising the one-hot feature tensor (which is
  754 GB per site at BASELINE config 4).

Reference quirk kept by default (``drop_last_channel=True``): the reference allocates
``max(ids)`` channels (jaxfeat.py:115), one fewer than the number of labels, so the
last-labelled constraint group gets no Gaussian features (Quirk A).  Labels are
the reference's own, order included (``constraint_group_labels`` reproduces id_feat's labels
bit for bit), so the group that loses its features is the one the reference drops.
"""
from typing import List, Optional, Tuple

import numpy as np

from .. import _kernels as K
from ..constraints import Constraints
from ..distributed import (agree_on_indices, agree_on_min, all_reduce_minmax_, all_reduce_sum_sym_, shard_extent,
                           take_global_frames)
from ..map import CLAFTMap, CLAMap, LinearMap
from .featlinearmap import KNAME_DIVS, KNAME_FEATS, KNAME_NAMES, constraint_group_labels, id_feat

DIVMETHOD_REORDER = "reorder"
DIVMETHOD_BASIC = "basic"
CLIP = 1e-3  # clipped_gauss default (jaxfeat.py:243-276)


def gb_centers(outer: float, inner: float, n_basis: int, dist_power: float, dtype=np.float32) -> np.ndarray:
    """Gaussian grid centres, uniform in r**dist_power (jaxfeat.py:235-236), in the feature dtype."""
    dt = np.dtype(dtype).type
    grid = np.linspace(inner**dist_power, outer**dist_power, n_basis).astype(dt)
    return (grid ** dt(1 / dist_power)).astype(dt)


def _feature_dtype(feature_dtype) -> np.dtype:
    dt = np.dtype(np.float32 if feature_dtype is None else feature_dtype)
    if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise TypeError(f"feature_dtype must be float32 or float64, not {dt}")
    return dt


class _Geometry:
    """Device-side per-trajectory quantities shared by all cg sites."""

    def __init__(self, coords, cmap: LinearMap, constraints: Constraints, drop_last_channel: bool,
                 feature_dtype=np.float32):
        import torch

        self.fdt = K.torch_dtype(_feature_dtype(feature_dtype))
        self.ids = constraint_group_labels(cmap.n_fg_sites, constraints)
        self.G = int(self.ids.max()) + 1
        order = np.argsort(self.ids, kind="stable").astype(np.int32)
        counts = np.bincount(self.ids, minlength=self.G)
        ptr = np.zeros(self.G + 1, dtype=np.int32)
        np.cumsum(counts, out=ptr[1:])
        c = K.as_device(coords)
        self.dev = c.device
        self.T = c.shape[0]
        self.grp_ptr = torch.from_numpy(ptr).to(self.dev)
        self.grp_atoms = torch.from_numpy(order).to(self.dev)
        self.sizes = torch.from_numpy(counts.astype(np.float32)).to(self.dev)
        self.n_ch = self.G - 1 if drop_last_channel else self.G
        # group-mean ("smeared") positions and mapped sites in the feature dtype: float32 like the reference's
        # JAX arrays unless feature_dtype=np.float64 was asked for
        self.Pg = K.group_reduce(c, self.grp_ptr, self.grp_atoms, self.G, True, self.fdt)
        self.cg = K.as_device(cmap(c)).to(self.fdt).contiguous()
        # group-summed coordinate map: sum_a M[c,a] [label(a) == g]
        M = np.asarray(cmap.standard_matrix, dtype=np.float64)
        self.Mg = np.add.reduceat(M[:, order], ptr[:-1], axis=1) if self.G < M.shape[1] else M[:, order]

    def group_forces(self, forces):
        f = K.as_device(forces)
        return K.group_reduce(f, self.grp_ptr, self.grp_atoms, self.G, False, f.dtype)


def gb_feat(
    points,
    cmap: LinearMap,
    constraints: Constraints,
    outer: float,
    inner: float = 0,
    n_basis: int = 10,
    width: float = 1.0,
    dist_power: float = 0.5,
    batch_size: Optional[int] = None,  # noqa: ARG001  (frames are streamed by the kernel)
    lazy: bool = True,
    div_method: str = DIVMETHOD_REORDER,
    drop_last_channel: bool = True,
    feature_dtype=np.float32,
):
    """Featurise each site by its distance to every mapped site (reference jaxfeat.py:20-184).

    Same arguments as the reference (``batch_size`` is accepted and unused; both ``div_method``
    values give the same closed-form divergence).  Returns the featuriser dictionary with
    per-site dense arrays: feats (n_frames, n_fg, n_basis*n_channels) and divs
    (n_frames, n_basis*n_channels, 3); generators if ``lazy``.  ``feature_dtype`` (no reference
    counterpart) is the arithmetic type of positions, distances and Gaussians: float32 is the
    reference's (JAX default) and the default; float64 evaluates the same expressions in double
    precision (the featurised fit amplifies float32 rounding of the features).
    """
    import torch

    if div_method not in (DIVMETHOD_REORDER, DIVMETHOD_BASIC):
        raise ValueError("Unknown method for jacobian calculation.")
    fdt = _feature_dtype(feature_dtype)
    geo = _Geometry(points, cmap, constraints, drop_last_channel, fdt)
    centers = torch.from_numpy(gb_centers(outer, inner, n_basis, dist_power, fdt)).to(geo.dev)
    ids_dev = torch.from_numpy(geo.ids.astype(np.int64)).to(geo.dev)
    keep = torch.nonzero(ids_dev < geo.n_ch).flatten()

    def site_arrays(site: int):
        gauss, grad = K.gb_channels(geo.Pg, geo.cg, site, geo.sizes, geo.n_ch, centers, width, CLIP)
        feats = torch.zeros((geo.T, cmap.n_fg_sites, geo.n_ch, n_basis), dtype=geo.fdt, device=geo.dev)
        if keep.numel():
            feats[:, keep, ids_dev[keep], :] = gauss[:, ids_dev[keep], :]
        return feats.reshape(geo.T, cmap.n_fg_sites, geo.n_ch * n_basis), grad.reshape(geo.T, geo.n_ch * n_basis, 3)

    def feat_of(site):
        return K.like_input(site_arrays(site)[0], points)

    def div_of(site):
        return K.like_input(site_arrays(site)[1], points)

    sites = range(cmap.n_cg_sites)
    if lazy:
        feats = (feat_of(s) for s in sites)
        divs = (div_of(s) for s in sites)
    else:
        feats = [feat_of(s) for s in sites]
        divs = [div_of(s) for s in sites]
    return {KNAME_FEATS: feats, KNAME_DIVS: divs, KNAME_NAMES: None}


# ----------------------------------------------------------------------------------------
# fused fit of the [id_feat | gb_feat] featuriser pair


def _bound_gb_kwargs(f) -> Optional[dict]:
    """kwargs of a gb_feat bound with Curry / util.curry / functools.partial, else None."""
    func = getattr(f, "func", None)
    if func is not gb_feat:
        return None
    if getattr(f, "args", ()):
        return None
    kw = dict(getattr(f, "kwargs", None) or getattr(f, "keywords", None) or {})
    allowed = {"outer", "inner", "n_basis", "width", "dist_power", "batch_size", "lazy", "div_method",
               "drop_last_channel", "feature_dtype"}
    if "outer" not in kw or not set(kw) <= allowed:
        return None
    return kw


def recognise(featurizers) -> Optional[Tuple[bool, Optional[dict]]]:
    """(use_id, gb kwargs or None) if the list is [id_feat], [gb] or [id_feat, gb]; else None."""
    fs = list(featurizers)
    if len(fs) == 1 and fs[0] is id_feat:
        return True, None
    if len(fs) == 1:
        kw = _bound_gb_kwargs(fs[0])
        return (False, kw) if kw is not None else None
    if len(fs) == 2 and fs[0] is id_feat:
        kw = _bound_gb_kwargs(fs[1])
        return (True, kw) if kw is not None else None
    return None


# Leave Gaussian columns that are identically zero over the trajectory out of the Gram matrix and the solve
# (their coefficients are exactly zero in the minimiser); tests switch it off to compare with the full system.
COMPACT_ZERO_COLUMNS = True
_SOLVE_MEMORY_FRACTION = 0.6  # share of the HBM not held by live tensors that a batch of sites may take


def _sites_per_batch(n_cg: int, n_feat: int, m: int, device) -> int:
    """How many cg sites are fitted side by side: each needs its Gram matrix, its constraint rows and one
    problem's share of the batched solver workspace.  288 GB of HBM hold all 64 sites of BASELINE config 4
    (n_feat 6139: 0.3 GB of Gram + 0.5 GB of workspace per site).  Derived from the device's TOTAL memory minus
    what live tensors hold (cached allocator blocks and the momentary free figure do not enter), so the value is
    the same from step to step; under ``comm=`` the ranks still agree on the minimum (``fit_id_gb``), because
    the batch size fixes the shape and the number of the all-reduces."""
    import torch

    per_site = 8 * (2 * n_feat * n_feat + m * n_feat + m) + K.eq_qp_batched_bytes(n_feat, m, 1, 1)
    _, total = torch.cuda.mem_get_info(device)
    usable = total - torch.cuda.memory_allocated(device)
    return int(max(1, min(n_cg, (_SOLVE_MEMORY_FRACTION * usable) // per_site)))


# Grouping of the sites for the batched solve, measured at BASELINE config 4 (tools/c4_batch_sweep.sh, solve ms per
# step): one batch of 64 sites 77.3; two of 32 64.9; four of 16 69.7; six of 8-16 78.0 -- below ~32 problems per
# launch the factorisation's kernels no longer fill the GPU.  Running a batch's solve on its own stream beside the
# next batch's Gram launches was tried and gains nothing (392.7 against 390.8 ms per step: the Gram kernel holds
# every CU's LDS, the solve's workgroups wait for its workgroups to retire and the sum stays the same).
_BATCH_MIN_SITES = 32
_BATCH_SIZE_RATIO = 0.85  # sites within 15 % of a batch's largest share its padding


def _solve_batches(n_act: List[int], per_batch: int, min_sites: int = _BATCH_MIN_SITES,
                   ratio: float = _BATCH_SIZE_RATIO) -> List[List[int]]:
    """Sites grouped for the batched solve: every problem of a batch is padded to the batch's largest, and the
    work of a solve grows with n^2..n^3, so sites are taken in order of decreasing size and a batch is closed
    when the next site is more than ``1 - ratio`` smaller than its first (once it holds ``min_sites``) or when
    it holds ``per_batch`` (the memory bound).  With a cut-off basis the sizes differ a lot between sites --
    BASELINE config 4: 1255..3049 kept columns, mean 2134 -- and one batch of all 64 sites padded to 3049 did
    2.6x the flops of the problems themselves.  Pure function of (n_act, per_batch): the same on every rank."""
    pad = lambda n: -(-n // 64) * 64
    order = sorted(range(len(n_act)), key=lambda i: (-n_act[i], i))
    out: List[List[int]] = []
    cur: List[int] = []
    for i in order:
        if cur and (len(cur) >= per_batch or (len(cur) >= min_sites and pad(n_act[i]) < ratio * pad(n_act[cur[0]]))):
            out.append(cur)
            cur = []
        cur.append(i)
    if cur:
        # a short last batch costs a whole chain of launches: it joins its predecessor when the memory allows
        if out and len(cur) < min_sites and len(out[-1]) + len(cur) <= per_batch:
            out[-1].extend(cur)
        else:
            out.append(cur)
    return out


def _fused_setup(coords, forces, coord_map: LinearMap, constraints: Constraints, use_id: bool,
                 gb_kwargs: Optional[dict], comm):
    """What the fused [id_feat | gb_feat] fit needs of a trajectory before any constraint frame is drawn: group
    geometry and group force sums on the device, the Gaussian centres, the feature counts, the group-summed
    coordinate map with its overlap ``Mg' Mg``, and per site the Gaussian columns that are not identically zero."""
    import types

    import torch

    kw = dict(gb_kwargs or {})
    drop_last = kw.pop("drop_last_channel", True)
    fdt = _feature_dtype(kw.pop("feature_dtype", np.float32))
    geo = _Geometry(coords, coord_map, constraints, drop_last, fdt)
    n_basis = int(kw.get("n_basis", 10)) if gb_kwargs is not None else 1
    width = float(kw.get("width", 1.0))
    centers_h = (gb_centers(kw["outer"], kw.get("inner", 0), n_basis, kw.get("dist_power", 0.5), fdt)
                 if gb_kwargs is not None else np.zeros(1, dtype=fdt))
    centers = torch.from_numpy(centers_h).to(geo.dev)
    n_id = geo.G if use_id else 0
    n_ch = geo.n_ch if gb_kwargs is not None else 0
    n_feat = n_id + n_ch * n_basis
    if n_feat == 0:
        raise ValueError("featuriser produces no features")
    Fg = geo.group_forces(forces)
    Mg = torch.from_numpy(np.ascontiguousarray(geo.Mg)).to(geo.dev)  # (n_cg, G) float64
    M2 = K.gb_group_overlap(Mg)  # Mg' Mg: every site's A'A is this, weighted (aggf_gb_constraint_gram)
    n_cg = coord_map.n_cg_sites
    # Which Gaussian columns can be non-zero at all?  Column (ch, k) of site c is identically zero when the
    # channel's distance to the site stays outside (c_k - h, c_k + h), h = width sqrt(ln(1/clip)), in every
    # frame (of every rank).  Such a column adds a zero row/column to P and zeros to A: its coefficient in
    # the minimiser is exactly 0, so it is left out (for a cut-off basis most columns are: BASELINE config 4
    # keeps ~2100 of 6139).  The distance range per (site, channel) is a superset test -- it can only keep
    # columns that are zero after all, never drop one that is not.
    keep = np.ones((n_cg, n_ch * n_basis), dtype=bool)
    if n_ch and COMPACT_ZERO_COLUMNS:
        rmin, rmax = K.gb_distance_range(geo.Pg, geo.cg, n_ch)
        all_reduce_minmax_(rmin, rmax, comm)
        lo = rmin.cpu().numpy()[:, :n_ch].astype(np.float64)
        hi = rmax.cpu().numpy()[:, :n_ch].astype(np.float64)
        reach = width * np.sqrt(np.log(1.0 / CLIP)) * (1.0 + 1e-5) + 1e-5  # float32 evaluation of the test: margin
        c = centers_h.astype(np.float64)[None, None, :]
        keep = ((lo[:, :, None] < c + reach) & (hi[:, :, None] > c - reach)).reshape(n_cg, n_ch * n_basis)
    cols_of = [np.nonzero(keep[site])[0].astype(np.int32) for site in range(n_cg)]
    n_act = [n_id + len(cols) for cols in cols_of]
    # Which of a site's variables can a constraint row touch at all?  Row (s, c) of A is Mg[c, g(f)] w_s(f): only
    # columns whose group lies in the support of the group-summed coordinate map -- for a slice map one group per cg
    # site, i.e. n_cg (1 + n_basis) variables of thousands.  The batched solve takes them LAST (_solve_order).
    hit = np.any(geo.Mg != 0.0, axis=0)  # (G,)
    touched = [np.concatenate([hit[:n_id], hit[cols // n_basis] if len(cols) else np.zeros(0, dtype=bool)])
               for cols in cols_of]
    return types.SimpleNamespace(touched=touched, geo=geo, fdt=fdt, drop_last=drop_last, n_basis=n_basis, width=width,
                                 centers_h=centers_h, centers=centers, n_id=n_id, n_ch=n_ch, n_feat=n_feat, Fg=Fg,
                                 Mg=Mg, M2=M2, n_cg=n_cg, cols_of=cols_of, n_act=n_act)


def _solve_order(sites: List[int], n_act: List[int], touched: List[np.ndarray], nb_max: int, device):
    """(perm (len(sites), nb_max) int32 on the device, a_first_col) for aggf_eq_qp_solve_batched_shift: per site the
    variables no constraint row touches first, then the batch's padding variables, then the touched ones;
    ``a_first_col`` = the number of leading variables untouched in EVERY site of the batch.  (None, 0) when that saves
    less than one 256-row block of the forward solve."""
    import os

    import torch

    first = nb_max - max(int(touched[i].sum()) for i in sites)
    if first < 256 or os.environ.get("AGGF_FEAT_ORDER", "1") == "0":  # (the switch: A/B measurements)
        return None, 0
    perm = np.empty((len(sites), nb_max), dtype=np.int32)
    for j, i in enumerate(sites):
        t = touched[i]
        idx = np.arange(n_act[i], dtype=np.int32)
        perm[j] = np.concatenate([idx[~t], np.arange(n_act[i], nb_max, dtype=np.int32), idx[t]])
    return torch.from_numpy(perm).to(device), int(first)


def fit_id_gb(
    traj,
    coord_map: LinearMap,
    kbt: float,
    n_constraint_frames: int,
    constraints: Constraints,
    l2_regularization: float,
    frame_indices,
    rng,
    comm,
    use_id: bool,
    gb_kwargs: Optional[dict],
    dense_featurizer,
) -> CLAFTMap:
    """qp_feat_linear_map (featlinearmap.py:249-394) for id_feat and/or gb_feat features, fused."""
    import torch

    su = _fused_setup(traj.coords, traj.forces, coord_map, constraints, use_id, gb_kwargs, comm)
    geo, fdt, drop_last, n_basis, width, centers_h, centers = (su.geo, su.fdt, su.drop_last, su.n_basis, su.width,
                                                                su.centers_h, su.centers)
    n_id, n_ch, n_feat, Fg, Mg, M2, n_cg = su.n_id, su.n_ch, su.n_feat, su.Fg, su.Mg, su.M2, su.n_cg
    cols_of, n_act = su.cols_of, su.n_act
    gen = np.random.default_rng() if rng is None else rng
    coefs: List[np.ndarray] = [None] * n_cg  # type: ignore [list-item]
    # Sampled constraint frames (featlinearmap.py:445): numbered over the WHOLE trajectory.  With frames
    # sharded over ranks every rank must build the same rows A, or the "replicated" solves differ: rank 0's
    # draw is used everywhere and each sampled frame's geometry comes from the rank that owns it.
    _, T_total = shard_extent(geo.T, comm, geo.dev)
    used: List[np.ndarray] = [
        np.asarray(frame_indices[site]) if frame_indices is not None
        else gen.choice(T_total, size=n_constraint_frames, replace=False)
        for site in range(n_cg)
    ]
    used = [agree_on_indices(idx, comm, geo.dev) for idx in used]
    flat_idx = np.concatenate(used) if used else np.zeros(0, dtype=np.int64)
    Pg_sel = take_global_frames(geo.Pg, flat_idx, comm)
    cg_sel = take_global_frames(geo.cg, flat_idx, comm)
    sel_begin = np.concatenate([[0], np.cumsum([len(u) for u in used])]).astype(np.int64)
    n_sel = {len(u) for u in used}
    n_max = max(n_act)
    # The regression matrix goes straight into the Gram kernel's in-place layout: float64 storage (the
    # float32 products of float32 forces widened on store -- K1 multiplies in float64, see below), feature
    # columns padded to a multiple of the 128-wide tile.  No pack pass, no float32 round trip.
    ld = -(-n_max // 128) * 128
    # The sites' K1 launches are short (4.5 ms at BASELINE config 4: ~140 tiles x a few frame ranges on 512 workgroup
    # slots) and end in a tail of half-empty CUs; dealt over a few streams, each with its own regression-matrix
    # buffer, the next site's workgroups fill the slots the previous launch leaves idle (AGGF_FEAT_STREAMS, default 3)
    import os

    n_str = max(1, min(8, int(os.environ.get("AGGF_FEAT_STREAMS", "3"))))
    free_b, _ = K.device_memory(geo.dev)
    while n_str > 1 and n_str * geo.T * ld * 24 > free_b // 4:  # one regression-matrix buffer per stream: never more
        n_str -= 1                                              # than a quarter of the free HBM between them
    main_stream = torch.cuda.current_stream(geo.dev)
    streams = K.side_streams(geo.dev, n_str) if n_str > 1 else [main_stream]
    R3s = [torch.zeros((geo.T, ld, 3), dtype=torch.float64, device=geo.dev) for _ in range(n_str)]
    lead_ready = None  # event: the first site's Gram matrix (the shared leading block) is complete
    # Sites are independent problems (own P, own A, one right-hand side): a batch of them is fitted side by
    # side -- K1 per site into one (sites, n_max, n_max) stack, ONE all-reduce of the stack, then ONE batched
    # K2 in which every step of the factorisation is a single launch over all sites.  (One solve alone is a
    # chain of ~300 small dependent kernels that leaves the GPU idle: 20 ms per site at n_feat = 6139.)
    # A site with fewer kept columns than n_max is padded with unit diagonal entries and zero constraint
    # columns: those variables come out as exact zeros.
    m_rows = max(n_sel) * n_cg if n_sel else 0
    per_batch = _sites_per_batch(n_cg, n_max, m_rows, geo.dev) if len(n_sel) == 1 else 1
    per_batch = agree_on_min(per_batch, comm, geo.dev)  # shapes and count of the collectives below depend on it
    shared_lead = None  # leading (id x id) Gram block, identical for all sites
    # every site's kept-column list in ONE upload (a pageable copy inside the site loop waits for its stream to drain:
    # 64 host stalls per fit)
    col_off = np.concatenate([[0], np.cumsum([len(c) for c in cols_of])]).astype(np.int64)
    cols_all = torch.from_numpy(np.concatenate(cols_of) if n_ch else np.zeros(0, dtype=np.int32)).to(geo.dev)
    cols_dev = [cols_all[int(col_off[i]):int(col_off[i + 1])] for i in range(n_cg)]
    batches = (_solve_batches(n_act, per_batch, int(os.environ.get("AGGF_FEAT_BATCH_MIN", _BATCH_MIN_SITES)),
                              float(os.environ.get("AGGF_FEAT_BATCH_RATIO", _BATCH_SIZE_RATIO)))
               if per_batch > 1 else [[i] for i in range(n_cg)])
    use_ata = os.environ.get("AGGF_FEAT_ATA", "1") != "0"
    for sites in batches:
        S = len(used[sites[0]])
        nb_max = max(n_act[i] for i in sites)  # this batch's padding
        Gs = torch.zeros((len(sites), nb_max, nb_max), dtype=torch.float64, device=geo.dev)
        As = torch.empty((len(sites), S * n_cg, nb_max), dtype=torch.float64, device=geo.dev)
        bs = torch.empty((len(sites), S * n_cg, 1), dtype=torch.float64, device=geo.dev)
        # A'A (the solve's positive shift) from the rows' structure, 20 multiply-adds per entry instead of 1280
        AtAs = torch.empty((len(sites), nb_max, nb_max), dtype=torch.float64, device=geo.dev)
        phase = K._timed("fit_sites")  # main-stream bracket of the whole site loop (the per-launch timers overlap)
        phase.__enter__()
        for st in streams:
            if st is not main_stream:
                st.wait_stream(main_stream)
        for j, site in enumerate(sites):
          with torch.cuda.stream(streams[j % n_str]):
            R3 = R3s[j % n_str]
            cols = cols_dev[site]
            na = n_act[site]
            K.gb_regmat_cols(Fg, geo.Pg, geo.cg, site, geo.sizes, n_id, cols, centers, width, CLIP, kbt, R3)
            # float64 products: with float32 products the Gram's rounding noise (~1e-7 of its largest entry)
            # exceeds l2 = 10 relative to force-squared sums of ~1e8 and P is no longer numerically positive
            # definite; the exact Gram of the float32 regression matrix always is
            # the id block of the regression matrix (the group force sums) is the same for every site: its
            # Gram block is taken from the first site's matrix (whole 128-tiles of it) and not computed again
            lead = (n_id // 128) * 128 if (shared_lead is not None and na >= 256) else 0
            if na == nb_max:
                K.gram(R3, None, None, na, torch.float64, out=Gs[j], first_col=lead)
                Gsite = Gs[j]
            else:
                Gsite = torch.empty((na, na), dtype=torch.float64, device=geo.dev)
                K.gram(R3, None, None, na, torch.float64, out=Gsite, first_col=lead)
            if lead:
                torch.cuda.current_stream(geo.dev).wait_event(lead_ready)
                Gsite[:lead, :lead] = shared_lead
            elif shared_lead is None and n_id >= 128:
                shared_lead = Gsite[: (n_id // 128) * 128, : (n_id // 128) * 128].clone()
                lead_ready = torch.cuda.Event()
                lead_ready.record(torch.cuda.current_stream(geo.dev))
            if na != nb_max:
                Gs[j, :na, :na] = Gsite
                Gs[j].diagonal()[na:] = 1.0
            lo_s, hi_s = int(sel_begin[site]), int(sel_begin[site + 1])
            gauss = None
            if n_ch:
                gauss, _ = K.gb_channels(Pg_sel[lo_s:hi_s].contiguous(), cg_sel[lo_s:hi_s].contiguous(), site,
                                         geo.sizes, n_ch, centers, width, CLIP)
            # (the constraint kernels stay inside the site loop: issued for all sites ahead of it they run alone,
            # 5 ms per step at BASELINE config 4 -- here they fill the tails of the Gram launches)
            K.gb_constraint_rows(Mg, gauss, S, n_id, n_ch, n_basis, site, out_A=As[j], out_b=bs[j], cols=cols)  # K4b
            if use_ata:
                K.gb_constraint_gram(M2, gauss, S, n_id, n_ch, n_basis, AtAs[j], cols=cols)
        for st in streams:
            if st is not main_stream:
                main_stream.wait_stream(st)
        phase.__exit__(None, None, None)
        all_reduce_sum_sym_(Gs, comm)
        perm, first = _solve_order(sites, n_act, su.touched, nb_max, geo.dev) if use_ata else (None, 0)
        X, stats = K.eq_qp_solve_batched(Gs, float(l2_regularization), None, As, bs, schur_reg=1e-12, n_refine=3,
                                         AtA=AtAs if use_ata else None, perm=perm, a_first_col=first)
        st_all = stats.cpu().numpy()
        X_host = X[:, 0, :].cpu().numpy()
        for j, site in enumerate(sites):
            st = st_all[j]
            if st[0] != 0 or not np.isfinite(st[1]):
                raise ValueError(
                    f"Map optimization failed. (site {site}: pivot {int(st[0])}, "
                    f"constraint residual {st[1]:.3e}, before refinement {st[2]:.3e}, scale {st[3]:.3e})"
                )
            full = np.zeros(n_feat, dtype=np.float64)
            full[:n_id] = X_host[j, :n_id]
            full[n_id + cols_of[site]] = X_host[j, n_id:n_act[site]]
            coefs[site] = full
        del Gs, As, bs, AtAs, X, stats
    # the batched solve's scratch (up to _SOLVE_MEMORY_FRACTION of the HBM) is not kept for the life of the
    # process: the next Gram / apply / streamed fit would find the memory gone
    K.drop_workspace("solve", geo.dev)
    fit_info = {"kept_columns": n_act, "n_feat": n_feat, "sites_per_batch": per_batch,
                "solve_batches": [(len(b), max(n_act[i] for i in b)) for b in batches],
                "kept_gauss_columns": cols_of, "feature_dtype": str(fdt)}
    coef_h = np.stack(coefs)
    # the application reads the coefficients as a compact list per site when most Gaussian columns were left out of the
    # fit (the cut-off basis: a lane per kept column instead of a lane per channel), as the dense matrix otherwise;
    # the list is the fit's own kept-column list -- no search for the non-zeros
    sparse = n_ch > 0 and sum(len(c) for c in cols_of) < 0.5 * n_cg * n_ch * n_basis
    compact = {}

    def compact_on(dev):
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        ptr_h = np.concatenate([[0], np.cumsum([len(c) for c in cols_of])]).astype(np.int32)
        vals = np.concatenate([coef_h[site, n_id + cols_of[site]] for site in range(n_cg)])
        return (to(coef_h[:, :n_id]) if n_id else None, to(ptr_h), to(np.concatenate(cols_of)), to(vals))

    def apply_f(points, copoints):
        g2 = _Geometry(copoints, coord_map, constraints, drop_last, fdt)
        key = str(g2.dev)
        if key not in compact:
            compact[key] = compact_on(g2.dev) if sparse else torch.from_numpy(coef_h).to(g2.dev)
        if sparse:
            out = K.gb_apply_cols(g2.group_forces(points), g2.Pg, g2.cg, g2.sizes, n_id, centers.to(g2.dev), width, CLIP,
                                  compact[key])
        else:
            out = K.gb_apply(g2.group_forces(points), g2.Pg, g2.cg, g2.sizes, n_id, n_ch, centers.to(g2.dev), width, CLIP,
                             compact[key])
        return K.like_input(out, points)

    from .featlinearmap import _feat_linear_mapping

    # only the scale/trans closures are borrowed (the reference's protocol for inspecting the map); the zero
    # test of CLAMap's constructor would run the DENSE featuriser once per site for nothing
    dense = _feat_linear_mapping(featurizer=dense_featurizer, coefs=coefs, mapping=coord_map,
                                 constraints=constraints, zeroes_check=False)
    force_map = CLAMap(scale=dense.scale, trans=dense.trans, n_fg_sites=coord_map.n_fg_sites,
                       n_cg_sites=n_cg, zeroes_check=False, apply=apply_f,
                       tags={"feat_names": None, "coef_list": coefs, "constraint_frames": used, "fit_info": fit_info})
    return CLAFTMap(coord_map=coord_map, force_map=force_map)


def cv_id_gb(coords, forces, coord_map: LinearMap, kbt: float, n_constraint_frames: int, constraints: Constraints,
             l2_values: List[float], folds: List[np.ndarray], rng, use_id: bool, gb_kwargs: Optional[dict]
             ) -> Optional[List[List[Optional[float]]]]:
    """Cross-validation of the fused [id_feat | gb_feat] fit over ``l2_regularization`` in ONE pass over the frames
    (the featurised counterpart of ``agg._grid_cv_gram_reuse``; replaces the loop body of the
    reference's agg.py:208-231 for this method).

    A site's Gram matrix is a sum over frames and the squared mapped hold-out force is a quadratic form in it: with the
    frames gathered into fold order once, every site's regression matrix is written once and K1 runs once per (site,
    fold); the training matrix of fold k is ``total - G_k`` and the hold-out score of the coefficients x_c is
    ``sum_c x_c' G_k^(c) x_c / (3 T_k n_cg)`` -- what ``force_smoothness`` of the mapped hold-out forces gives.  Per
    (grid point, fold) only the constraint rows (drawn anew, as a fresh ``qp_feat_linear_map`` call would: the
    generator is consumed in the order of the reference's loop) and one batched solve are left.  The Gaussian columns
    kept are those not identically zero over ALL frames, a superset of every training set's (their extra coefficients
    are exact zeros: needs l2 > 0, which the caller checks).

    Returns ``scores[i][k]`` for l2_values[i] and fold k (None where the solve failed, as the loop would skip it), or
    None -- before anything is drawn from ``rng`` -- when the per-fold matrices would not fit the device."""
    import torch

    n_folds = len(folds)
    lens = [len(f) for f in folds]
    bounds = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    c_dev, f_dev = K.as_device(coords), K.as_device(forces)
    pidx = torch.as_tensor(np.concatenate(folds), device=c_dev.device)
    c_p, f_p = K.take_frames(c_dev, pidx), K.take_frames(f_dev, pidx)  # frames in fold order: a fold is a row range
    su = _fused_setup(c_p, f_p, coord_map, constraints, use_id, gb_kwargs, None)
    del c_p, f_p
    geo, n_id, n_ch, n_basis, n_cg = su.geo, su.n_id, su.n_ch, su.n_basis, su.n_cg
    cols_of, n_act = su.cols_of, su.n_act
    dev = geo.dev
    n_max = max(n_act)
    ld = -(-n_max // 128) * 128
    # every site keeps its per-fold matrices (twice unless kbt = 1) for the whole grid: BASELINE config 4 holds 23 GB of
    # them; a system whose matrices do not fit beside the regression matrix and a batch of solves takes the loop
    same = float(kbt) == 1.0
    held = (1 if same else 2) * (n_folds + 1) * 8 * sum(n * n for n in n_act) + geo.T * ld * 24
    free_b, _ = K.device_memory(dev)
    if held > 0.5 * free_b:
        return None
    cols_dev = [torch.from_numpy(c).to(dev) for c in cols_of]
    R3 = torch.zeros((geo.T, ld, 3), dtype=torch.float64, device=dev)
    lead_n = (n_id // 128) * 128
    shared_lead: List[Optional[torch.Tensor]] = [None] * n_folds  # the id x id block is the same for every site
    # The fit minimises |feat'F + kbt div|^2 (featlinearmap.py:361-369), the fitted map adds the divergence term with
    # weight ONE (trans_f, featlinearmap.py:512-520): the hold-out score is a quadratic form in the Gram matrix of
    # feat'F + div, the training matrix comes from feat'F + kbt div -- two regression matrices per site unless kbt = 1
    # (the id x id block holds no divergence: it is shared between the two and between the sites)
    fold_grams, score_grams, totals = [], [], []
    for site in range(n_cg):
        na = n_act[site]
        per_alpha = []
        for alpha in ((float(kbt),) if same else (float(kbt), 1.0)):
            K.gb_regmat_cols(su.Fg, geo.Pg, geo.cg, site, geo.sizes, n_id, cols_dev[site], su.centers, su.width, CLIP,
                             alpha, R3)
            Gf = torch.empty((n_folds, na, na), dtype=torch.float64, device=dev)
            for k in range(n_folds):
                lead = lead_n if (shared_lead[k] is not None and na >= 256) else 0
                K.gram(R3[int(bounds[k]):int(bounds[k + 1])], None, None, na, torch.float64, out=Gf[k], first_col=lead)
                if lead:
                    Gf[k][:lead, :lead] = shared_lead[k]
                elif shared_lead[k] is None and lead_n >= 128:
                    shared_lead[k] = Gf[k][:lead_n, :lead_n].clone()
            per_alpha.append(Gf)
        tot = per_alpha[0][0].clone()
        for k in range(1, n_folds):
            K.axpby(1.0, tot, 1.0, per_alpha[0][k], out=tot)
        fold_grams.append(per_alpha[0])
        score_grams.append(per_alpha[-1])
        totals.append(tot)
    del R3
    gen = np.random.default_rng() if rng is None else rng
    S = int(n_constraint_frames)
    per_batch = _sites_per_batch(n_cg, n_max, S * n_cg, dev)
    batches = _solve_batches(n_act, per_batch) if per_batch > 1 else [[i] for i in range(n_cg)]
    orders = [_solve_order(b, n_act, su.touched, max(n_act[i] for i in b), dev) for b in batches]
    scores: List[List[Optional[float]]] = []
    for l2 in l2_values:
        row: List[Optional[float]] = []
        for k in range(n_folds):
            # the frames a fit on the training subset would draw, numbered in the subset (= fold order without fold k)
            used = [gen.choice(geo.T - lens[k], size=S, replace=False) for _ in range(n_cg)]
            flat = np.concatenate([np.where(u < bounds[k], u, u + lens[k]) for u in used])
            sel = torch.as_tensor(flat, device=dev)
            Pg_sel, cg_sel = geo.Pg[sel].contiguous(), geo.cg[sel].contiguous()
            total_q, ok = 0.0, True
            for sites, (perm, first) in zip(batches, orders):
                nb_max = max(n_act[i] for i in sites)
                Gs = torch.zeros((len(sites), nb_max, nb_max), dtype=torch.float64, device=dev)
                As = torch.empty((len(sites), S * n_cg, nb_max), dtype=torch.float64, device=dev)
                bs = torch.empty((len(sites), S * n_cg, 1), dtype=torch.float64, device=dev)
                AtAs = torch.empty((len(sites), nb_max, nb_max), dtype=torch.float64, device=dev)
                for j, site in enumerate(sites):
                    na = n_act[site]
                    if na == nb_max:
                        K.axpby(1.0, totals[site], -1.0, fold_grams[site][k], out=Gs[j])
                    else:
                        Gs[j, :na, :na] = K.axpby(1.0, totals[site], -1.0, fold_grams[site][k])
                        Gs[j].diagonal()[na:] = 1.0
                    gauss = None
                    if n_ch:
                        gauss, _ = K.gb_channels(Pg_sel[site * S:(site + 1) * S], cg_sel[site * S:(site + 1) * S], site,
                                                 geo.sizes, n_ch, su.centers, su.width, CLIP)
                    K.gb_constraint_rows(su.Mg, gauss, S, n_id, n_ch, n_basis, site, out_A=As[j], out_b=bs[j],
                                         cols=cols_dev[site])
                    K.gb_constraint_gram(su.M2, gauss, S, n_id, n_ch, n_basis, AtAs[j], cols=cols_dev[site])
                X, stats = K.eq_qp_solve_batched(Gs, float(l2), None, As, bs, schur_reg=1e-12, n_refine=3, AtA=AtAs,
                                                 perm=perm, a_first_col=first)
                st_all = stats.cpu().numpy()
                bad = [(sites[j], st_all[j]) for j in range(len(sites))
                       if st_all[j][0] != 0 or not np.isfinite(st_all[j][1])]
                if bad:
                    site, st = bad[0]
                    print(f"Map optimization failed. (site {site}: pivot {int(st[0])}, constraint residual {st[1]:.3e}, "
                          f"before refinement {st[2]:.3e}, scale {st[3]:.3e})")
                    ok = False
                    break
                q = [K.gram_quadform(score_grams[site][k], X[j, :, :n_act[site]].contiguous()) for j, site in enumerate(sites)]
                total_q += float(torch.cat(q).sum().item())
                del Gs, As, bs, AtAs, X, stats
            row.append(total_q / (3.0 * lens[k] * n_cg) if ok else None)
        scores.append(row)
    K.drop_workspace("solve", dev)
    return scores
