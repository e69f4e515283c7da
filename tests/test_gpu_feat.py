"""GPU: gb_feat (K4) and the featurised fit against the CPU oracle.  gb_feat is parity-unpinned
(the reference needs JAX); the oracle restates jaxfeat.py and is self-checked by finite
differences in tests/test_oracle_golden.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from aggforce_amd import LinearMap, Trajectory, project_forces  # noqa: E402
from aggforce_amd.qp import Multifeaturize, gb_feat, id_feat, qp_feat_linear_map  # noqa: E402
from aggforce_amd.util import Curry  # noqa: E402
from oracle import aggforce_oracle as orc  # noqa: E402

KBT = 0.6955215


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def system(T=40, N=14, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    coords = (6 * rng.random((T, N, 3)) + 1).astype(dtype)
    forces = (25 * rng.standard_normal((T, N, 3))).astype(dtype)
    cons = {frozenset([1, 2]), frozenset([4, 5]), frozenset([5, 6]), frozenset([10, 13])}
    # two-atom sites: no cg site coincides with a (smeared) atom, so r > 0 everywhere (at r == 0 the
    # reference's norm gradient is NaN and poisons the whole site)
    cmat = orc.list_mapping_matrix([[0, 1], [4, 7], [8, 10], [12, 13]], N)
    return coords, forces, cons, cmat


def oracle_feats(coords, cmat, cons, ids, n_channels, **kw):
    N = coords.shape[1]
    smear = orc.smear_matrix(orc.reduce_constraint_sets(cons), N) if cons else np.eye(N, dtype=np.float32)
    cg = orc.linearmap_apply(coords, cmat)
    out = [orc.gb_feat_site(coords, cg[:, c, :], ids, smear, n_channels=n_channels, **kw)
           for c in range(cmat.shape[0])]
    return [o[0] for o in out], [o[1] for o in out]


@pytest.mark.parametrize("with_cons", [True, False])
@pytest.mark.parametrize("drop_last", [True, False])
def test_gb_feat_dense_matches_oracle(with_cons, drop_last):
    coords, forces, cons, cmat = system()
    if not with_cons:
        cons = set()
    cmap = LinearMap(cmat)
    ids = orc.id_feat_ids(coords.shape[1], cons)  # the reference's labels: decides the dropped channel
    kw = dict(outer=8.0, inner=0.0, n_basis=5, width=1.0, dist_power=0.5)
    res = gb_feat(coords, cmap, cons, lazy=False, drop_last_channel=drop_last, **kw)
    n_ch = int(ids.max()) + (0 if drop_last else 1)
    of, od = oracle_feats(coords, cmat, cons, ids, n_ch, **kw)
    assert res["names"] is None and len(res["feats"]) == 4
    for c in range(4):
        f, d = res["feats"][c], res["divs"][c]
        assert f.dtype == np.float32 and f.shape == of[c].shape and d.shape == od[c].shape
        assert np.max(np.abs(f - of[c])) < 2e-6
        assert np.isfinite(od[c]).all() and np.isfinite(d).all()
        assert np.max(np.abs(d - od[c])) < 2e-5
    lazy = gb_feat(coords, cmap, cons, drop_last_channel=drop_last, **kw)
    assert rel(next(iter(lazy["feats"])), of[0]) < 1e-5


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_feat_fit_matches_dense_path_and_oracle(dtype):
    coords, forces, cons, cmat = system(T=60, dtype=dtype)
    cmap = LinearMap(cmat)
    kw = dict(outer=8.0, inner=0.0, n_basis=4, width=1.0)
    feat = Multifeaturize([id_feat, Curry(gb_feat, **kw)])
    rng = np.random.default_rng(3)
    frames = [rng.choice(60, size=6, replace=False) for _ in range(4)]
    traj = Trajectory(coords=coords, forces=forces)
    fused = qp_feat_linear_map(traj, cmap, feat, KBT, constraints=cons, frame_indices=frames, l2_regularization=10.0)
    dense = qp_feat_linear_map(traj, cmap, feat, KBT, constraints=cons, frame_indices=frames, l2_regularization=10.0,
                               fused=False)
    cf = np.stack(fused.force_map.tags["coef_list"])
    cd = np.stack(dense.force_map.tags["coef_list"])
    assert rel(cf, cd) < 2e-4
    mf, md = fused(traj), dense(traj)
    assert rel(mf.forces, md.forces) < 2e-4 and rel(mf.coords, md.coords) < 1e-6
    # oracle: dense float32 features in the reference's label order, exact solve, with the SAME six sampled frames
    # per site.  The Gram matrix of the oracle is formed in float64 here (forces widened): a float32 Gram, which is
    # what featlinearmap.py:361-370 produces for float32 arrays, carries rounding noise of the order of l2 and moves
    # the optimum by 1e-3 with the summation order alone (profiles/r03_feat_conditioning.txt); the product forms
    # the exact Gram of its float32 regression matrix.  20 frames per site: tests/test_gpu_feat20.py.
    ids = orc.id_feat_ids(coords.shape[1], cons)  # the reference's labels: decides the dropped channel
    G = int(ids.max()) + 1
    onehot = np.zeros((60, coords.shape[1], G), dtype=np.float32)
    onehot[:, np.arange(coords.shape[1]), ids] = 1
    gf, gd = oracle_feats(coords.astype(np.float32), cmat, cons, ids, G - 1, dist_power=0.5, **kw)
    feats = [np.concatenate([onehot, g], axis=2) for g in gf]
    divs = [np.concatenate([np.zeros((60, G, 3), np.float32), d], axis=1) for d in gd]
    ocoef = orc.qp_feat_linear_map(forces.astype(np.float64), cmat, feats, divs, KBT, frames, 10.0)
    assert rel(cf, np.stack(ocoef)) < 1e-3
    assert rel(mf.forces, orc.cla_apply(forces.astype(np.float64), feats, divs, ocoef)) < 1e-3
    # piecewise: regression matrix and Gram of site 1 against the oracle's
    from aggforce_amd import _kernels as K
    from aggforce_amd.qp.gbfeat import CLIP, _Geometry, gb_centers

    geo = _Geometry(coords, cmap, cons, True)
    Fg = geo.group_forces(forces)
    n_feat = G + 4 * (G - 1)
    R3 = torch.zeros((60, 128, 3), dtype=Fg.dtype, device="cuda")
    K.gb_regmat(Fg, geo.Pg, geo.cg, 1, geo.sizes, G, G - 1, torch.from_numpy(gb_centers(8.0, 0.0, 4, 0.5)).cuda(), 1.0,
                CLIP, KBT, R3)
    reg_o, qp_o = orc.feat_site_problem(forces, feats[1], divs[1], KBT, 0.0)
    assert rel(np.swapaxes(R3.cpu().numpy()[:, :n_feat, :], 1, 2).reshape(-1, n_feat), reg_o) < 2e-6
    assert rel(K.gram(R3, None, None, n_feat, R3.dtype).cpu().numpy(), qp_o) < 1e-5
    # constraint rows are satisfied on the sampled frames: sum_a W_c(t)[a] M[c',a] = delta_cc'
    scale = fused.force_map.scale(coords[frames[1]])
    assert np.max(np.abs(np.einsum("tca,da->tcd", scale, cmat)[:, 1, :] - np.eye(4)[1])) < 1e-5
    # through project_forces, gb only and id only
    res = project_forces(coords, forces, cmap, constrained_inds=cons, method=qp_feat_linear_map,
                         featurizer=Multifeaturize([Curry(gb_feat, **kw)]), kbt=KBT, frame_indices=frames)
    assert res["mapped_forces"].shape == (60, 4, 3) and np.isfinite(res["residual"])
    only_id = qp_feat_linear_map(traj, cmap, Multifeaturize([id_feat]), KBT, constraints=cons, frame_indices=frames)
    ref_id = qp_feat_linear_map(traj, cmap, id_feat, KBT, constraints=cons, frame_indices=frames)
    assert rel(only_id(traj).forces, ref_id(traj).forces) < 2e-4


def test_fused_fit_larger_system_runs_and_reduces_residual():
    """512 atoms, 16 sites, n_basis 8: the dense feature tensor would be 512*4104*T floats per
    site; the fused path never forms it.  Featurised maps must not be worse than the linear map
    on the training set (their feature space contains the linear maps' id features)."""
    rng = np.random.default_rng(11)
    T, N, n_cg = 1500, 512, 16
    base = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3) * 1.5
    coords = (base[None] + 0.3 * rng.standard_normal((T, N, 3))).astype(np.float32)
    forces = (30 * rng.standard_normal((T, N, 3))).astype(np.float32)
    forces += 3.0 * (coords - coords.mean(axis=1, keepdims=True))  # configuration-dependent part
    cmap = LinearMap([[i * 32, i * 32 + 5] for i in range(n_cg)], n_fg_sites=N)
    cons = {frozenset([3 * i, 3 * i + 1]) for i in range(0, 60)}
    feat = Multifeaturize([id_feat, Curry(gb_feat, outer=8.0, n_basis=8)])
    res = project_forces(coords, forces, cmap, constrained_inds=cons, method=qp_feat_linear_map, featurizer=feat,
                         kbt=KBT, rng=np.random.default_rng(0), l2_regularization=10.0)
    lin = project_forces(coords, forces, cmap, constrained_inds=cons, l2_regularization=0.0)
    assert np.isfinite(res["residual"]) and res["residual"] < lin["residual"] * 1.02
    assert len(res["tmap"].force_map.tags["coef_list"]) == n_cg


def test_gb_feat_random_sweep_matches_oracle():
    """gb_feat features and divergences (K4) for random geometries, constraint sets, bases and flags."""
    rng = np.random.default_rng(77)
    worst_f = worst_d = 0.0
    for case in range(30):
        N = int(rng.integers(6, 28))
        T = int(rng.integers(1, 25))
        dtype = rng.choice([np.float32, np.float64])
        coords = (7 * rng.random((T, N, 3)) + 1).astype(dtype)
        cons = set()
        for _ in range(int(rng.integers(0, N // 2))):
            size = int(rng.choice([2, 2, 3]))
            cons.add(frozenset(int(i) for i in rng.choice(N, size=size, replace=False)))
        n_cg = int(rng.integers(1, 5))
        # two-atom sites with distinct weights: a site never coincides with a (smeared) atom position
        cmat = np.zeros((n_cg, N))
        for c in range(n_cg):
            i, j = rng.choice(N, size=2, replace=False)
            cmat[c, i], cmat[c, j] = 0.37, 0.63
        cmap = LinearMap(cmat)
        ids = orc.id_feat_ids(N, cons)
        drop_last = bool(rng.random() < 0.5)
        if drop_last and int(ids.max()) == 0:
            drop_last = False  # a single group with the last channel dropped leaves no channels
        kw = dict(outer=float(rng.choice([6.0, 9.0])), inner=float(rng.choice([0.0, 1.5])),
                  n_basis=int(rng.integers(1, 7)), width=float(rng.choice([0.7, 1.0, 1.8])),
                  dist_power=float(rng.choice([0.5, 1.0])))
        res = gb_feat(coords, cmap, cons, lazy=False, drop_last_channel=drop_last, **kw)
        n_ch = int(ids.max()) + (0 if drop_last else 1)
        of, od = oracle_feats(coords, cmat, cons, ids, n_ch, **kw)
        for c in range(n_cg):
            f, d = res["feats"][c], res["divs"][c]
            assert f.shape == of[c].shape and d.shape == od[c].shape, (case, f.shape, of[c].shape)
            if not (np.isfinite(of[c]).all() and np.isfinite(od[c]).all()):
                continue  # r == 0 by coincidence: the reference's NaN
            worst_f = max(worst_f, float(np.max(np.abs(f - of[c]))))
            worst_d = max(worst_d, float(np.max(np.abs(d - od[c]))))
    assert worst_f < 5e-6 and worst_d < 1e-4, (worst_f, worst_d)


def test_gb_feat_kernels_match_autodiff_fixture(golden):
    """K4 against the autodiff derivation of tests/golden/g7 (see oracle/gen_g7_autodiff.py): the transcribed
    forward pass of jaxfeat.py differentiated by torch.autograd, labels from the reference's id_feat."""
    from conftest import cons_in_insertion_order

    g = golden("g7_gbfeat_autodiff.npz")
    for name in [str(n) for n in g["names"]]:
        outer, inner, n_basis, width, dist_power = g[f"{name}__kw"]
        coords, cmat = g[f"{name}__coords"], g[f"{name}__cmat"]
        cons = cons_in_insertion_order(g[f"{name}__cons"])
        cmap = LinearMap(cmat)
        assert np.array_equal(id_feat(coords, cmap, cons, return_ids=True), g[f"{name}__ids"])
        for method in ("reorder", "basic"):
            res = gb_feat(coords, cmap, cons, outer=float(outer), inner=float(inner), n_basis=int(n_basis),
                          width=float(width), dist_power=float(dist_power), lazy=False, div_method=method)
            ref_d = g[f"{name}__divs"] if method == "reorder" else g[f"{name}__divs_basic"]
            for c in range(cmat.shape[0]):
                f, d = res["feats"][c], res["divs"][c]
                assert f.shape == g[f"{name}__feats"][c].shape and d.shape == ref_d[c].shape
                assert np.max(np.abs(f - g[f"{name}__feats"][c])) < 5e-6, (name, c)
                assert np.max(np.abs(d - ref_d[c])) < 1e-4, (name, c, method)


def test_zero_column_compaction_is_exact():
    """Gaussian columns that vanish over the whole trajectory (channels beyond the cut-off of a site) are left out
    of the Gram matrix and the solve; the fit must equal the full system's: same coefficients (exact zeros at
    the dropped columns), same mapped forces, and the oracle's solution of the FULL dense problem."""
    from aggforce_amd.qp import gbfeat

    rng = np.random.default_rng(21)
    T, N = 80, 24
    # two clusters 40 apart: every atom of one cluster is beyond outer + reach of the sites in the other
    base = np.concatenate([3.0 * rng.random((N // 2, 3)), 3.0 * rng.random((N // 2, 3)) + 40.0])
    coords = (base[None] + 0.2 * rng.standard_normal((T, N, 3))).astype(np.float32)
    forces = (25 * rng.standard_normal((T, N, 3))).astype(np.float32)
    cons = {frozenset([1, 2]), frozenset([13, 14]), frozenset([14, 15]), frozenset([20, 23])}
    cmat = orc.list_mapping_matrix([[0, 1], [4, 7], [12, 13], [18, 22]], N)
    cmap = LinearMap(cmat)
    kw = dict(outer=6.0, inner=0.0, n_basis=5, width=1.0)
    feat = Multifeaturize([id_feat, Curry(gb_feat, **kw)])
    frames = [np.array([k]) for k in (3, 17, 40, 66)]
    traj = Trajectory(coords=coords, forces=forces)
    assert gbfeat.COMPACT_ZERO_COLUMNS
    small = qp_feat_linear_map(traj, cmap, feat, KBT, constraints=cons, frame_indices=frames, l2_regularization=10.0)
    info = small.force_map.tags["fit_info"]
    assert max(info["kept_columns"]) < 0.7 * info["n_feat"]  # the far cluster's columns are gone
    try:
        gbfeat.COMPACT_ZERO_COLUMNS = False
        full = qp_feat_linear_map(traj, cmap, feat, KBT, constraints=cons, frame_indices=frames, l2_regularization=10.0)
    finally:
        gbfeat.COMPACT_ZERO_COLUMNS = True
    assert full.force_map.tags["fit_info"]["kept_columns"] == [info["n_feat"]] * 4
    cs, cf = np.stack(small.force_map.tags["coef_list"]), np.stack(full.force_map.tags["coef_list"])
    assert cs.shape == cf.shape == (4, info["n_feat"])
    assert rel(cs, cf) < 1e-7
    assert rel(small(traj).forces, full(traj).forces) < 1e-7
    # dropped columns: exact zeros, and the dense features really vanish there in every frame
    ids = orc.id_feat_ids(N, cons)
    G = int(ids.max()) + 1
    gf, gd = oracle_feats(coords, cmat, cons, ids, G - 1, dist_power=0.5, **kw)
    for c in range(4):
        dead = np.abs(gf[c]).max(axis=(0, 1)) == 0
        assert dead.sum() > 0 and np.all(cs[c, G:][dead] == 0.0)
    # oracle on the full dense problem
    onehot = np.zeros((T, N, G), dtype=np.float32)
    onehot[:, np.arange(N), ids] = 1
    feats = [np.concatenate([onehot, g], axis=2) for g in gf]
    divs = [np.concatenate([np.zeros((T, G, 3), np.float32), d], axis=1) for d in gd]
    # (Gram in float64, like the product's: the float32 Gram of featlinearmap.py:361-370 is only reproducible to
    # ~1e-3 -- round 2 compared with it and had to loosen this to 1e-2 / 2e-3; profiles/r03_feat_conditioning.txt)
    ocoef = orc.qp_feat_linear_map(forces.astype(np.float64), cmat, feats, divs, KBT, frames, 10.0)
    assert rel(cs, np.stack(ocoef)) < 1e-3
    assert rel(small(traj).forces, orc.cla_apply(forces.astype(np.float64), feats, divs, ocoef)) < 1e-3


@pytest.mark.parametrize("with_id", [True, False])
def test_featurised_grid_cv_one_pass_matches_the_loop(with_id):
    """project_forces_grid_cv over l2_regularization of the fused featurised fit: the one-pass form (per-fold Gram
    matrices of every site, training matrix = total - fold, hold-out score as a quadratic form) against the
    reference's loop of fits and applications, fed by identically seeded generators (the one-pass form draws the
    constraint frames in the loop's order).  The two score the hold-out frames through different arithmetic (float64
    quadratic form of the float64 Gram matrix / float32 features applied frame by frame): 1e-5."""
    from aggforce_amd import agg

    coords, forces, cons, cmat = system(T=240, seed=21)
    cmap = LinearMap(cmat)
    gb = Curry(gb_feat, outer=8.0, inner=0.0, n_basis=4, width=1.0)
    feat = Multifeaturize([id_feat, gb] if with_id else [gb])
    grid = {"l2_regularization": [0.5, 10.0, 300.0]}
    calls = {"n": 0}
    real = agg._grid_cv_feat_reuse

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    def go(reuse):
        # method_rng is the method's generator (constraint frames); the rng argument shuffles the folds
        return agg.project_forces_grid_cv(grid, coords, forces, n_folds=4, rng=np.random.default_rng(3),
                                          reuse_gram=reuse, method_rng=np.random.default_rng(17), coord_map=cmap,
                                          constrained_inds=cons, method=qp_feat_linear_map, featurizer=feat, kbt=KBT,
                                          n_constraint_frames=6)

    agg._grid_cv_feat_reuse = counted
    try:
        fast, loop = go(True), go(False)
    finally:
        agg._grid_cv_feat_reuse = real
    assert calls["n"] == 1  # the one-pass form really ran, and only for reuse_gram=True
    assert set(fast) == {"scores", "sds", "n_runs"}
    for key in loop["scores"]:
        assert fast["n_runs"][key] == loop["n_runs"][key] == 4
        assert abs(fast["scores"][key] - loop["scores"][key]) < 1e-5 * abs(loop["scores"][key]), key
        assert abs(fast["sds"][key] - loop["sds"][key]) < 1e-3 * abs(loop["sds"][key]) + 1e-6 * abs(loop["scores"][key]), key
    vals = [loop["scores"][k] for k in loop["scores"]]
    assert len(set(np.round(vals, 6))) == 3  # the grid points really differ


@pytest.mark.parametrize("fdt", [np.float32, np.float64])
@pytest.mark.parametrize("n_id_on", [True, False])
def test_gb_apply_from_compact_coefficients_matches_dense(fdt, n_id_on):
    """aggf_gb_apply_cols (one lane per non-zero Gaussian coefficient) against aggf_gb_apply (one lane per channel,
    zeros skipped) on the same coefficients: ragged frame count, sites with no Gaussian coefficient at all, more basis
    functions than the dense kernel holds in registers."""
    from aggforce_amd import _kernels as K
    from aggforce_amd.qp.gbfeat import CLIP, _Geometry, gb_centers

    rng = np.random.default_rng(77)
    T, N, n_cg, nb = 203, 90, 6, 11
    coords = (9 * rng.random((T, N, 3)) + 1).astype(np.float32)
    forces = (20 * rng.standard_normal((T, N, 3))).astype(np.float32)
    cons = {frozenset([3 * i, 3 * i + 1]) for i in range(20)}
    cmap = LinearMap([[7 * i, 7 * i + 3] for i in range(n_cg)], n_fg_sites=N)
    geo = _Geometry(coords, cmap, cons, True, fdt)
    Fg = geo.group_forces(forces)
    n_id = geo.G if n_id_on else 0
    n_ch = geo.n_ch
    centers = torch.from_numpy(gb_centers(8.0, 0.0, nb, 0.5, fdt)).cuda()
    coef = rng.standard_normal((n_cg, n_id + n_ch * nb)) * (rng.random((n_cg, n_id + n_ch * nb)) < 0.2)
    coef[3, n_id:] = 0.0  # a site whose Gaussian block is empty
    dense = K.gb_apply(Fg, geo.Pg, geo.cg, geo.sizes, n_id, n_ch, centers, 1.0, CLIP, torch.from_numpy(coef).cuda())
    compact = K.gb_compact_coefficients(coef, n_id, geo.dev)
    assert int(compact[1][-1]) == np.count_nonzero(coef[:, n_id:])
    got = K.gb_apply_cols(Fg, geo.Pg, geo.cg, geo.sizes, n_id, centers, 1.0, CLIP, compact)
    assert got.shape == (T, n_cg, 3)
    # the two kernels form bit-identical terms (the feature helpers forbid fused multiply-add contraction, which would
    # otherwise be the compiler's choice per kernel: 2e-7 apart in float32); what is left is the order of the float64 sums
    assert rel(got.cpu().numpy(), dense.cpu().numpy()) < 1e-12


def test_featurised_grid_cv_falls_back_to_the_loop_when_the_matrices_do_not_fit(monkeypatch):
    """The one-pass form keeps (folds, n, n) per site for the whole grid; when that does not fit the device it declines
    BEFORE drawing from the method's generator, and the loop -- fed by the untouched generator -- gives exactly what
    reuse_gram=False gives."""
    from aggforce_amd import agg
    from aggforce_amd.qp import gbfeat

    coords, forces, cons, cmat = system(T=120, seed=5)
    cmap = LinearMap(cmat)
    feat = Multifeaturize([id_feat, Curry(gb_feat, outer=8.0, inner=0.0, n_basis=4, width=1.0)])
    grid = {"l2_regularization": [1.0, 10.0]}

    def go(reuse):
        return agg.project_forces_grid_cv(grid, coords, forces, n_folds=3, rng=np.random.default_rng(3), reuse_gram=reuse,
                                          method_rng=np.random.default_rng(9), coord_map=cmap, constrained_inds=cons,
                                          method=qp_feat_linear_map, featurizer=feat, kbt=KBT, n_constraint_frames=5)

    declined = {"n": 0}
    real = agg._grid_cv_feat_reuse

    def watched(*a, **k):
        out = real(*a, **k)
        declined["n"] += out is None
        return out

    monkeypatch.setattr(agg, "_grid_cv_feat_reuse", watched)
    monkeypatch.setattr(gbfeat.K, "device_memory", lambda dev=None: (1 << 16, 1 << 16))
    tight = go(True)
    assert declined["n"] == 1
    monkeypatch.undo()
    loop = go(False)
    for key in loop["scores"]:
        assert tight["scores"][key] == loop["scores"][key] and tight["n_runs"][key] == loop["n_runs"][key] == 3


def test_fused_fit_is_independent_of_the_solve_order_and_batching(monkeypatch):
    """The batched solve of the fused fit may take the variables the sparse constraint rows touch last
    (gbfeat._solve_order: aggf_eq_qp_solve_batched_shift with perm / a_first_col), form A'A from the rows' structure
    and group the sites by size: each of these is an exact reformulation.  A system large enough for all three to be
    active (slice map, ~1000 kept columns per site) against the same fit with them switched off."""
    from aggforce_amd.qp import gbfeat

    rng = np.random.default_rng(13)
    T, N, n_cg = 400, 330, 8
    grid = np.stack(np.meshgrid(*[np.arange(7)] * 3, indexing="ij"), -1).reshape(-1, 3)[:N] * 1.6
    coords = (grid[None] + 0.25 * rng.standard_normal((T, N, 3))).astype(np.float32)
    forces = (25 * rng.standard_normal((T, N, 3))).astype(np.float32)
    cons = {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
    cmap = LinearMap([[3 * (i * (N // n_cg) // 3)] for i in range(n_cg)], n_fg_sites=N)
    feat = Multifeaturize([id_feat, Curry(gb_feat, outer=9.0, inner=0.0, n_basis=6, width=1.0)])
    traj = Trajectory(coords=coords, forces=forces)
    frames = [rng.choice(T, size=12, replace=False) for _ in range(n_cg)]
    seen = {}
    real_order = gbfeat._solve_order

    def spy(*a, **k):
        out = real_order(*a, **k)
        seen["first"] = out[1]
        return out

    monkeypatch.setattr(gbfeat, "_solve_order", spy)
    monkeypatch.setattr(gbfeat, "_BATCH_MIN_SITES", 4)
    a = qp_feat_linear_map(traj, cmap, feat, KBT, constraints=cons, frame_indices=frames, l2_regularization=5.0)
    info = a.force_map.tags["fit_info"]
    assert seen["first"] >= 256 and min(info["kept_columns"]) > 500  # the restricted forward solve really ran
    monkeypatch.setenv("AGGF_FEAT_ORDER", "0")
    monkeypatch.setenv("AGGF_FEAT_ATA", "0")
    monkeypatch.setenv("AGGF_FEAT_BATCH_MIN", "64")
    b = qp_feat_linear_map(traj, cmap, feat, KBT, constraints=cons, frame_indices=frames, l2_regularization=5.0)
    assert len(b.force_map.tags["fit_info"]["solve_batches"]) == 1
    ca, cb = np.stack(a.force_map.tags["coef_list"]), np.stack(b.force_map.tags["coef_list"])
    assert rel(ca, cb) < 1e-9, rel(ca, cb)
    assert rel(a(traj).forces, b(traj).forces) < 1e-9
