"""CPU: the NumPy oracle against the golden fixtures produced by the reference itself
(oracle/gen_golden.py) and against the reference's own test data."""
import numpy as np
import pytest

from oracle import aggforce_oracle as orc
from conftest import cons_from_array


def rel(a, b):
    return np.max(np.abs(np.asarray(a, float) - np.asarray(b, float))) / max(1.0, np.max(np.abs(b)))


def test_waterdimer_known_answer(golden):
    """tests/test_agg.py:17-44 of the reference: optimal map ~ per-molecule aggregation."""
    g = golden("g1_waterdimer.npz")
    W = orc.qp_linear_map(g["forces"], g["coord_matrix"], set())
    assert np.allclose(W, g["known_answer"], atol=5e-3)
    # SURVEY 8(c) probe values of the exact optimum
    assert abs(W[0, 1] - 1.000291) < 2e-6 and abs(W[1, 4] - 1.004972) < 2e-6


@pytest.mark.parametrize("cname", ["none", "guessed"])
@pytest.mark.parametrize("l2", [0.0, 1.0, 1e3])
def test_waterdimer_golden(golden, cname, l2):
    g = golden("g1_waterdimer.npz")
    cons = set() if cname == "none" else cons_from_array(g["guessed_constraints"])
    key = f"{cname}_l2_{l2:g}"
    pr = orc.linear_problem(g["forces"], g["coord_matrix"], cons, l2)
    assert rel(pr["qp_mat"], g[f"{key}__qp_mat"]) < 1e-12
    assert np.array_equal(pr["con_mat"], g[f"{key}__con_mat"])
    assert rel(pr["A"], g[f"{key}__A"]) < 1e-14
    res = orc.project_forces(g["coords"], g["forces"], g["coord_matrix"], cons, l2)
    assert rel(res["force_map"], g[f"{key}__W"]) < 1e-8
    assert rel(res["mapped_forces"], g[f"{key}__mapped_forces"]) < 1e-8
    assert abs(res["residual"] - float(g[f"{key}__residual"])) < 1e-8 * float(g[f"{key}__residual"])
    # constraint residual of the exact solve
    assert np.max(np.abs(g["coord_matrix"] @ res["force_map"].T - np.eye(2))) < 1e-10


def test_guess_constraints_waterdimer(golden):
    g = golden("g1_waterdimer.npz")
    assert orc.guess_pairwise_constraints(g["coords"][:10]) == cons_from_array(g["guessed_constraints"])


@pytest.mark.parametrize("mname", ["slice", "dense", "block"])
@pytest.mark.parametrize("cname", ["none", "pairs", "chain", "overlap"])
def test_synthetic_linear_golden(golden, mname, cname):
    g = golden("g2_synthetic_linear.npz")
    cons = cons_from_array(g[f"cons_{cname}"]) if g[f"cons_{cname}"].size else set()
    cmat = g[f"map_{mname}"]
    for dt in ("float64", "float32"):
        for l2 in (0.0, 2.5):
            key = f"{mname}_{cname}_{dt}_l2_{l2:g}"
            f = g["forces"].astype(dt)
            c = g["coords"].astype(dt)
            res = orc.project_forces(c, f, cmat, cons, l2)
            assert rel(res["force_map"], g[f"{key}__W"]) < 1e-7
            assert rel(res["mapped_forces"], g[f"{key}__mapped_forces"]) < 1e-7
            assert rel(res["mapped_coords"], g[f"{key}__mapped_coords"]) < 1e-12


def test_linearmap_semantics_golden(golden):
    g = golden("g3_linearmap.npz")
    assert rel(orc.linearmap_apply(g["pos"], g["mat"]), g["call"]) < 1e-13
    assert np.array_equal(orc.list_mapping_matrix([[0, 2, 3], [4]], 6), g["list_ctor"])
    assert rel(orc.linearmap_apply(g["nan_pos"], g["nan_mat"]), g["nan_call"]) < 1e-13
    with pytest.raises(ValueError):
        orc.linearmap_apply(g["nan_bad_pos"], g["nan_mat"])
    off = orc.linearmap_apply(g["nan_bad_pos"], g["nan_mat"], handle_nans=False)
    assert np.array_equal(np.isnan(off), np.isnan(g["nan_off_call"]))


def test_cln025_saved_maps(golden):
    """tests/test_forces.py:132-185: saved basic map reproduced exactly; structural invariants
    of the saved optimum (its trajectory is absent from the mount)."""
    g = golden("g4_cln025.npz")
    pairs = cons_from_array(g["pairs"])
    W = orc.constraint_aware_uni_map(g["coord_matrix"], pairs)
    assert ((W - g["basic"]) ** 2).sum() < 1e-5
    C = orc.make_bond_constraint_matrix(int(g["n_atoms"]), pairs)
    assert np.array_equal(C, g["con_mat"]) and C.shape == (175, 97)
    opt = g["opt"]
    assert np.max(np.abs(g["coord_matrix"] @ opt.T - np.eye(10))) < 1e-12  # M W' = I
    x = np.linalg.lstsq(C, opt.T, rcond=None)[0]
    assert np.max(np.abs(C @ x - opt.T)) < 1e-12  # W in range(C)


def test_feat_id_golden(golden):
    g = golden("g5_feat_id.npz")
    cons = cons_from_array(g["cons"])
    ids = orc.id_feat_ids(12, cons)
    assert np.array_equal(ids, g["ids"])
    feats, divs = orc.id_feat(48, 12, cons)
    for l2 in (10.0, 0.5):
        frames = list(g[f"l2_{l2:g}__frames"])
        coefs = orc.qp_feat_linear_map(g["forces"], g["coord_matrix"], [feats] * 4, [divs] * 4, float(g["kbt"]),
                                       frames, l2)
        assert rel(np.stack(coefs), g[f"l2_{l2:g}__coefs"]) < 1e-7
        mapped = orc.cla_apply(g["forces"], [feats] * 4, [divs] * 4, coefs)
        assert rel(mapped, g[f"l2_{l2:g}__mapped_forces"]) < 1e-6
    assert np.array_equal(orc.smear_matrix(orc.reduce_constraint_sets(cons), 12), g["smear"])


def test_augmented_golden(golden):
    g = golden("g6_augmented.npz")
    cons = cons_from_array(g["cons"])
    oc, of = orc.augment(g["coords"], g["forces"], g["coord_matrix"], float(g["var"]), float(g["kbt"]), g["eps_fit"])
    assert rel(oc, g["aug_coords"]) < 1e-6 and rel(of, g["aug_forces"]) < 1e-6
    o = orc.joptgauss_force_map(g["coords"], g["forces"], g["coord_matrix"], float(g["var"]), float(g["kbt"]),
                                g["eps_fit"], cons)
    assert rel(o["force_map"], g["W"]) < 1e-6
    ds, dg = orc.condnormal_log_gradient(g["scn_src"], g["scn_gen"], np.eye(5), 0.3)
    assert np.allclose(ds, g["scn_dsrc"], atol=2e-6) and np.allclose(dg, g["scn_dgen"], atol=2e-6)


def test_gb_feat_divergence_matches_finite_differences():
    """gb_feat is parity-unpinned (JAX absent): self-check the closed-form divergence against
    central differences of the summed features, in float64."""
    rng = np.random.default_rng(7)
    T, N = 3, 7
    pts = 3 * rng.random((T, N, 3)) + 1.0
    cons = {frozenset([1, 2]), frozenset([4, 5, 6])}
    ids = orc.id_feat_ids(N, cons)
    smear = orc.smear_matrix(orc.reduce_constraint_sets(cons), N).astype(np.float64)
    cg = pts[:, 0, :] + 0.37
    kw = dict(outer=6.0, inner=0.0, n_basis=4, width=1.0, dist_power=0.5, dtype=np.float64)
    nch = int(ids.max()) + 1
    feats, divs = orc.gb_feat_site(pts, cg, ids, smear, n_channels=nch, **kw)
    assert feats.shape == (T, N, 4 * nch) and divs.shape == (T, 4 * nch, 3)
    h = 1e-6
    num = np.zeros_like(divs)
    for a in range(N):
        for d in range(3):
            p1, p2 = pts.copy(), pts.copy()
            p1[:, a, d] += h
            p2[:, a, d] -= h
            f1, _ = orc.gb_feat_site(p1, cg, ids, smear, n_channels=nch, **kw)
            f2, _ = orc.gb_feat_site(p2, cg, ids, smear, n_channels=nch, **kw)
            # d/dx_a of sum over atoms a' of feature columns; attribute to the channel of atom a
            dsum = (f1.sum(axis=1) - f2.sum(axis=1)) / (2 * h)  # (T, n_feat): all channels' sums
            # jacrev(sum_{t,a'} gauss)[k, t, a, d] then channel_allocate puts it in channel(a)
            g1 = np.zeros((T, 4))
            for ch in range(nch):
                g1 += dsum[:, 4 * ch:4 * ch + 4]
            num[:, 4 * ids[a]:4 * ids[a] + 4, d] += g1
    assert np.max(np.abs(num - divs)) < 1e-5 * max(1.0, np.max(np.abs(divs)))
    # quirk A: default n_channels = max(ids) drops the last label's channel
    f_q, d_q = orc.gb_feat_site(pts, cg, ids, smear, **kw)
    assert f_q.shape[2] == 4 * (nch - 1)
    assert np.array_equal(f_q, feats[:, :, : 4 * (nch - 1)])


def _g7_cases(g):
    for name in [str(n) for n in g["names"]]:
        outer, inner, n_basis, width, dist_power = g[f"{name}__kw"]
        kw = dict(outer=float(outer), inner=float(inner), n_basis=int(n_basis), width=float(width),
                  dist_power=float(dist_power))
        yield name, kw


def test_gb_feat_oracle_matches_autodiff_fixture(golden):
    """Second derivation (oracle/gen_g7_autodiff.py): the forward pass of jaxfeat.py transcribed op by op and
    differentiated AUTOMATICALLY (torch.autograd; 'reorder' and 'basic' methods of gb_subfeat_jac) against the
    oracle's hand-derived closed form -- multi-atom groups, the clip boundary, the dropped last channel."""
    from conftest import cons_in_insertion_order

    g = golden("g7_gbfeat_autodiff.npz")
    for name, kw in _g7_cases(g):
        coords, cmat = g[f"{name}__coords"], g[f"{name}__cmat"]
        cons = cons_in_insertion_order(g[f"{name}__cons"])
        N = coords.shape[1]
        ids = orc.id_feat_ids(N, cons)
        assert np.array_equal(ids, g[f"{name}__ids"])  # the reference's labels
        smear = orc.smear_matrix(orc.reduce_constraint_sets(cons), N) if cons else np.eye(N, dtype=np.float32)
        cg = orc.linearmap_apply(coords, cmat.astype(np.float32))
        for c in range(cmat.shape[0]):
            f, d = orc.gb_feat_site(coords, cg[:, c, :], ids, smear, **kw)
            rf, rd, rb = g[f"{name}__feats"][c], g[f"{name}__divs"][c], g[f"{name}__divs_basic"][c]
            assert f.shape == rf.shape and d.shape == rd.shape
            assert f.shape[2] == kw["n_basis"] * int(ids.max())  # Quirk A: max(ids) channels, the last label has none
            assert np.max(np.abs(f - rf)) < 5e-6, name
            assert np.max(np.abs(d - rd)) < 5e-5 and np.max(np.abs(d - rb)) < 5e-5, name
    # the clip boundary case really straddles the boundary: the atom just inside has a small positive
    # feature and a non-zero gradient, the one just outside exactly zero
    f = g["clip_edge__feats"][0]
    ids = g["clip_edge__ids"]
    nb, checked = 2, 0
    if ids[2] < ids.max():  # (the last label has no channel)
        inside = f[:, 2, nb * ids[2]]
        assert np.all(inside > 0) and np.all(inside < 1e-3)
        assert np.all(np.abs(g["clip_edge__divs"][0][:, nb * ids[2], 0]) > 1e-4)
        checked += 1
    if ids[3] < ids.max():
        assert np.all(f[:, 3, nb * ids[3]] == 0) and np.all(g["clip_edge__divs"][0][:, nb * ids[3], :] == 0)
        checked += 1
    assert checked >= 1


def test_exact_featurised_problem_is_ill_posed_when_float32_flips_the_rank():
    """Why the featurised fit at 20 constraint frames is compared in float64 features, and in float32 only where
    the rank is stable (tests/test_gpu_feat20.py): when a cg site is the midpoint of two unconstrained atoms, both
    are exactly equidistant from it, their Gaussian rows coincide and the 20 n_cg constraint rows lose rank in exact
    arithmetic.  float32 rounding makes the rows independent again at the 1e-7 level; the EXACT equality-constrained
    optimum then enforces those rounding-level rows and moves by O(0.1) -- for the oracle itself, float32 against
    float64 features, with everything else in float64.  No implementation can be within 1e-3 of both."""
    from oracle.feat_cases import KBT, L2, dense_features, geometry, numerical_rank

    coords, forces, cons, cmat, kw, frames = geometry("box14_degenerate")
    f64, d64 = dense_features(coords, cmat, cons, kw, np.float64)
    f32, d32 = dense_features(coords.astype(np.float32), cmat, cons, kw, np.float32)
    x64 = orc.qp_feat_linear_map(forces, cmat, f64, d64, KBT, frames, L2)
    wide = [f.astype(np.float64) for f in f32], [d.astype(np.float64) for d in d32]
    x32 = orc.qp_feat_linear_map(forces, cmat, wide[0], wide[1], KBT, frames, L2)   # float32 FEATURES, float64 arithmetic
    shift = []
    for c in range(cmat.shape[0]):
        r64, _ = orc.feat_site_problem(forces, f64[c], d64[c], KBT, 0.0)
        r32, _ = orc.feat_site_problem(forces, wide[0][c], wide[1][c], KBT, 0.0)
        a, b = r32 @ x32[c], r64 @ x64[c]
        A64, _ = orc.feat_constraint_arrays(f64[c], c, cmat, frames[c])
        A32, _ = orc.feat_constraint_arrays(f32[c], c, cmat, frames[c])
        shift.append((numerical_rank(A32) - numerical_rank(A64), float(np.max(np.abs(a - b)) / np.max(np.abs(b)))))
    assert [s[0] for s in shift] == [0, 0, shift[2][0], 0] and shift[2][0] > 0      # only the midpoint site flips
    assert shift[2][1] > 1e-2                                                         # ... and its optimum moves
    assert max(shift[0][1], shift[1][1], shift[3][1]) < 1e-5                          # the others do not


def test_full_covariance_log_gradient_is_the_gradient_of_scipys_logpdf():
    """``orc.condnormal_full_log_gradient`` (closed form of jaxgausstraj.py:77-96) against central differences of
    scipy's multivariate-normal log-density: an anchor independent of our algebra (JAX itself is absent)."""
    from scipy.stats import multivariate_normal

    rng = np.random.default_rng(8)
    N, n = 5, 2
    M = rng.standard_normal((n, N))
    B = rng.standard_normal((3 * n, 3 * n))
    cov = B @ B.T + 0.5 * np.eye(3 * n)
    x = rng.standard_normal((1, N, 3))
    y = orc.trjdot(x, M) + 0.3 * rng.standard_normal((1, n, 3))

    def logp(xx, yy):
        return multivariate_normal.logpdf(yy.reshape(-1), mean=orc.trjdot(xx, M).reshape(-1), cov=cov)

    d_src, d_gen = orc.condnormal_full_log_gradient(x, y, M, cov)
    h = 1e-5
    for arr, grad, wrt in ((x, d_src, 0), (y, d_gen, 1)):
        num = np.zeros_like(arr)
        for i in np.ndindex(arr.shape):
            p, m = arr.copy(), arr.copy()
            p[i] += h
            m[i] -= h
            num[i] = ((logp(p, y) - logp(m, y)) if wrt == 0 else (logp(x, p) - logp(x, m))) / (2 * h)
        assert np.max(np.abs(num - grad)) < 1e-6 * max(1.0, np.max(np.abs(grad)))
    # and a sample with injected noise has mean premap(x) and the Cholesky factor applied to the noise
    eps = rng.standard_normal((1, n, 3))
    ys = orc.condnormal_full_sample(x, M, cov, eps)
    L = np.linalg.cholesky(cov)
    assert np.allclose(ys.reshape(-1), orc.trjdot(x, M).reshape(-1) + L @ eps.reshape(-1))
