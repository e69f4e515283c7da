"""GPU: the dtype / mode / tile variants of the kernels behind one entry point, each against the oracle (or NumPy's
promotion of the same expression).  tests/test_gpu_zz_coverage.py reported, the first time it ran (round 5), that a
third of the library's template instantiations were launched by no test: every `NAN_REPLACE` form of the apply
kernels, float64-in / float32-out applies, four of five `residual_over_var` variants, half of the gb_feat kernels'
dtype triples.  This file names each of them once."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from aggforce_amd import LinearMap, _lib  # noqa: E402
from aggforce_amd import _kernels as K  # noqa: E402
from oracle import aggforce_oracle as orc  # noqa: E402

TD = {np.float32: torch.float32, np.float64: torch.float64}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def launched(family):
    """Demangled names of the kernels of `family` this process has launched since the last reset."""
    return sorted(p.split("(")[0].replace("void ", "") for p, c in _lib.coverage(names=True).values() if family in p and c > 0)


# ------------------------------------------------------------------ K3: every tile x dtype pair x NaN mode
# (n_cg, N, T): few sites with 8 / 12 / 20 / 40 frames per stage (by frame size, per input dtype), 32- / 48- / 64-site
# tiles, the 128-site 16-wave tile (with a float64 map and 32 atoms or more the plain mode of these goes to the LDS-DMA
# kernels -- tests/test_gpu_parity.py::test_apply_wide_tile_nan_scan_and_ragged_tiles -- so each tile also gets a shape
# with fewer than 32 atoms, which keeps it on the register-staged kernel)
APPLY_SHAPES = [(10, 175, 5003), (4, 20, 100003), (5, 64, 5003), (16, 40, 5001), (16, 97, 5003), (7, 130, 5003), (3, 300, 5003),
                (20, 77, 333), (40, 200, 257), (60, 130, 300), (130, 150, 200), (200, 1001, 129),
                (20, 24, 333), (40, 28, 257), (60, 30, 300), (130, 31, 200)]


@pytest.mark.parametrize("pdt,mdt", [(np.float64, np.float64), (np.float32, np.float64), (np.float32, np.float32),
                                     (np.float64, np.float32)])
def test_apply_every_tile_in_both_nan_modes(pdt, mdt):
    """aggf_linearmap_apply (map/core.py:219-240, util.py:119-125) for every (points dtype, map dtype) pair the C ABI
    takes -- float64 points with a float32 map is what CondNormal's premap of float64 coordinates in a float32
    augmenter asks for (jaxgausstraj.py:232: the source is cast to the augmenter's dtype first) -- in the plain mode and
    with NaN replacement (the two fills of the reference's NaN policy, map/core.py:222-231), fused sum of squares and
    NaN probe included."""
    rng = np.random.default_rng(5)
    _lib.load().aggf_coverage_reset()
    for n_cg, N, T in APPLY_SHAPES:
        pts = (50 * rng.standard_normal((T, N, 3))).astype(pdt)
        mat = rng.standard_normal((n_cg, N)).astype(mdt)
        p, m = dev(pts), dev(mat)
        tol = 1e-13 if (pdt == np.float64 and mdt == np.float64) else 3e-5
        ref = orc.trjdot(pts.astype(mdt), mat) if mdt == np.float32 else orc.trjdot(pts, mat)
        probe = torch.zeros(1, dtype=torch.int32, device="cuda")
        out, ss = K.linearmap_apply(p, m, want_sumsq=True, nan_probe=probe)
        assert out.dtype == TD[mdt] and rel(out.cpu().numpy(), ref) < tol and int(probe.item()) == 0, (n_cg, N, T)
        want_ss = float((ref.astype(np.float64) ** 2).sum())
        assert abs(float(ss.item()) - want_ss) < 1e-4 * want_ss
        # NaN replacement: NaNs read as the fill, the probe reports them
        bad = pts.copy()
        holes = rng.integers(0, T, size=9), rng.integers(0, N, size=9), rng.integers(0, 3, size=9)
        bad[holes] = np.nan
        for fill in (0.0, -1.0):
            filled = pts.copy()
            filled[holes] = fill
            want = orc.trjdot(filled.astype(mdt), mat) if mdt == np.float32 else orc.trjdot(filled, mat)
            probe.zero_()
            got, ss2 = K.linearmap_apply(dev(bad), m, nan_fill=fill, want_sumsq=True, nan_probe=probe)
            assert rel(got.cpu().numpy(), want) < tol and int(probe.item()) == 1, (n_cg, N, T, fill)
            assert abs(float(ss2.item()) - float((want.astype(np.float64) ** 2).sum())) < 1e-4 * want_ss
    names = launched("apply_")
    tin, tc = ("double" if pdt == np.float64 else "float"), ("double" if mdt == np.float64 else "float")
    for kb, nblk in ((8, 2), (12, 4), (20, 4), (40, 8)):
        for mode in ("false", "true"):
            assert f"aggf::apply_small_kernel<{tin}, {tc}, {mode}, {kb}, {nblk}>" in names, (kb, mode, names)
    for threads, tile in ((256, 32), (256, 48), (256, 64), (1024, 128)):
        for mode in ("false", "true"):
            assert f"aggf::apply_kernel<{tin}, {tc}, {mode}, {threads}, {tile}, 4>" in names, (tile, mode, names)


# ------------------------------------------------------------------ K5: dtype variants
@pytest.mark.parametrize("gdt,mdt,odt", [(np.float64, np.float64, np.float64), (np.float64, np.float32, np.float64),
                                         (np.float32, np.float64, np.float64), (np.float32, np.float32, np.float64),
                                         (np.float32, np.float32, np.float32)])
def test_residual_over_var_dtype_variants(gdt, mdt, odt):
    """(g - mean) / var and its negative (jaxgausstraj.py:263-284 in closed form) in every input / output dtype the
    C ABI takes, against NumPy evaluated in the output dtype."""
    rng = np.random.default_rng(3)
    g = (5 * rng.standard_normal((1237, 7, 3))).astype(gdt)
    mean = (5 * rng.standard_normal((1237, 7, 3))).astype(mdt)
    pos, neg = K.residual_over_var(dev(g), dev(mean), 0.37, TD[odt])
    want = (g.astype(odt) - mean.astype(odt)) / odt(0.37)
    tol = 1e-15 if odt == np.float64 else 1e-6
    assert pos.dtype == TD[odt] and rel(pos.cpu().numpy(), want) < tol and rel(neg.cpu().numpy(), -want) < tol
    only_neg = K.residual_over_var(dev(g), dev(mean), 0.37, TD[odt], want_pos=False)
    assert only_neg[0] is None and torch.equal(only_neg[1], neg)


@pytest.mark.parametrize("cdt,adt", [(np.float64, np.float64), (np.float32, np.float64), (np.float64, np.float32),
                                     (np.float32, np.float32)])
def test_condnormal_augment_and_sites_dtype_variants(cdt, adt):
    """The fused K5 pass (trajectory/core.py:382-390 with JCondNormal, jaxgausstraj.py:213-289) for every (trajectory
    dtype, augmenter dtype) pair, with injected noise, against the oracle's `augment`; `condnormal_sites` gives the
    generated sites of the same pass."""
    rng = np.random.default_rng(8)
    T, N, n_cg, var, kbt = 517, 12, 3, 0.05, 0.6955215
    coords = (5 * rng.random((T, N, 3))).astype(cdt)
    forces = (30 * rng.standard_normal((T, N, 3))).astype(cdt)
    M = orc.list_mapping_matrix([[0, 1], [4], [7, 8, 9]], N)
    noise = rng.standard_normal((T, n_cg, 3)).astype(adt)
    odt = np.result_type(cdt, adt).type
    want_c, want_f = orc.augment(coords.astype(odt), forces.astype(odt), M, var, kbt, noise, dtype=odt)
    c, f = dev(coords), dev(forces)
    cols = K.premap_columns(M.astype(adt), TD[adt], c.device)
    mean = K.linearmap_apply(c, dev(M.astype(adt)))
    oc, of = K.condnormal_augment(c, f, cols, n_cg, mean, var, kbt, dev(noise), 1, 0)
    tol = 1e-12 if odt == np.float64 and adt == np.float64 and cdt == np.float64 else 2e-4
    assert oc.dtype == TD[odt] and rel(oc.cpu().numpy(), want_c) < tol and rel(of.cpu().numpy(), want_f) < tol
    y, fa = K.condnormal_sites(mean, var, kbt, dev(noise), 1, 0, TD[odt])
    assert rel(y.cpu().numpy(), want_c[:, N:]) < tol and rel(fa.cpu().numpy(), want_f[:, N:]) < tol


@pytest.mark.parametrize("cdt,gdt", [(np.float64, np.float64), (np.float32, np.float64), (np.float64, np.float32),
                                     (np.float32, np.float32)])
def test_augment_concat_dtype_variants(cdt, gdt):
    """The general Augmenter protocol's concatenation (trajectory/core.py:384-390) for every (trajectory dtype,
    augmenter dtype) pair against NumPy."""
    rng = np.random.default_rng(4)
    T, N, n = 333, 9, 4
    x, F = rng.standard_normal((T, N, 3)).astype(cdt), rng.standard_normal((T, N, 3)).astype(cdt)
    y, corr, lg = (rng.standard_normal(s).astype(gdt) for s in ((T, n, 3), (T, N, 3), (T, n, 3)))
    kbt = 0.7
    oc, of = K.augment_concat(dev(x), dev(F), dev(y), dev(corr), dev(lg), kbt)
    want_c = np.concatenate([x, y], axis=1)
    want_f = np.concatenate([F + kbt * corr, kbt * lg], axis=1)
    assert oc.dtype == TD[want_c.dtype.type] and np.array_equal(oc.cpu().numpy(), want_c)
    assert of.dtype == TD[want_f.dtype.type]
    assert rel(of.cpu().numpy(), want_f) < (1e-15 if cdt == gdt == np.float64 else 1e-6)


# ------------------------------------------------------------------ K4: dtype triples of the gb_feat kernels
@pytest.mark.parametrize("tf,tg,to", [(np.float32, np.float32, np.float32), (np.float32, np.float32, np.float64),
                                      (np.float64, np.float32, np.float64), (np.float32, np.float64, np.float64),
                                      (np.float64, np.float64, np.float64)])
def test_gb_regression_matrix_and_apply_dtype_triples(tf, tg, to):
    """aggf_gb_regmat / aggf_gb_regmat_cols / aggf_gb_apply / aggf_gb_apply_cols (featlinearmap.py:361-369 and 512-520
    on jaxfeat.py's features) for every (forces, geometry, output) dtype triple the C ABI takes: the float64 triple
    against the oracle's dense formulation, the others against the float64 kernels on the same (rounded) inputs."""
    from aggforce_amd.qp.gbfeat import CLIP, _Geometry, gb_centers

    rng = np.random.default_rng(21)
    T, N, n_cg, nb, kbt = 91, 30, 4, 5, 0.6955215
    coords = (6 * rng.random((T, N, 3)) + 1).astype(np.float32)
    forces = (25 * rng.standard_normal((T, N, 3))).astype(np.float32)
    cons = {frozenset([3 * i, 3 * i + 1]) for i in range(6)}
    cmat = orc.list_mapping_matrix([[0, 2], [7, 8], [13, 14], [22, 29]], N)
    cmap = LinearMap(cmat)
    geo = {np.float32: _Geometry(coords, cmap, cons, True, np.float32), np.float64: _Geometry(coords, cmap, cons, True, np.float64)}
    G, n_ch = geo[tg].G, geo[tg].n_ch
    n_feat = G + nb * n_ch
    ld = 256

    def run(tf_, tg_, to_):
        g = geo[tg_]
        Fg = g.group_forces(forces).to(TD[tf_])
        centers = torch.from_numpy(gb_centers(8.0, 0.0, nb, 0.5, tg_)).cuda()
        R = torch.zeros((T, ld, 3), dtype=TD[to_], device="cuda")
        K.gb_regmat(Fg, g.Pg, g.cg, 2, g.sizes, G, n_ch, centers, 1.0, CLIP, kbt, R)
        keep = torch.arange(1, nb * n_ch, 2, dtype=torch.int32, device="cuda")  # every other Gaussian column
        Rc = torch.zeros((T, ld, 3), dtype=TD[to_], device="cuda")
        K.gb_regmat_cols(Fg, g.Pg, g.cg, 2, g.sizes, G, keep, centers, 1.0, CLIP, kbt, Rc)
        coef = np.random.default_rng(1).standard_normal((n_cg, n_feat)) * (np.random.default_rng(2).random((n_cg, n_feat)) < 0.4)
        dense = K.gb_apply(Fg, g.Pg, g.cg, g.sizes, G, n_ch, centers, 1.0, CLIP, torch.from_numpy(coef).cuda())
        compact = K.gb_apply_cols(Fg, g.Pg, g.cg, g.sizes, G, centers, 1.0, CLIP, K.gb_compact_coefficients(coef, G, g.dev))
        return R.cpu().numpy(), Rc.cpu().numpy(), keep.cpu().numpy(), dense.cpu().numpy(), compact.cpu().numpy()

    R, Rc, keep, dense, compact = run(tf, tg, to)
    R64, Rc64, _, dense64, _ = run(np.float64, np.float64, np.float64)
    tol = 1e-12 if (tf, tg, to) == (np.float64,) * 3 else 3e-5
    assert R.dtype == to and rel(R[:, :n_feat], R64[:, :n_feat]) < tol and float(np.abs(R[:, n_feat:]).max()) == 0.0
    # the compact form: [id columns | listed Gaussian columns] of the full matrix
    same = 1e-14 if tg == np.float64 else 1e-6  # (float32 features: the two kernels round their products differently)
    assert rel(Rc[:, :G], R[:, :G]) < same and rel(Rc[:, G:G + len(keep)], R[:, G + keep]) < same
    assert float(np.abs(Rc[:, G + len(keep):]).max()) == 0.0
    assert rel(dense, dense64) < tol and rel(compact, dense) < 1e-12
    if (tf, tg, to) == (np.float64,) * 3:
        ids = orc.id_feat_ids(N, cons)
        smear = orc.smear_matrix(orc.reduce_constraint_sets(cons), N)
        cg = orc.linearmap_apply(coords, cmat)
        gf, gd = orc.gb_feat_site(coords, cg[:, 2, :], ids, smear, outer=8.0, inner=0.0, n_basis=nb, width=1.0, dist_power=0.5,
                                  n_channels=G - 1)
        onehot = np.zeros((T, N, G), dtype=np.float32)
        onehot[:, np.arange(N), ids] = 1
        feat = np.concatenate([onehot, gf], axis=2)
        div = np.concatenate([np.zeros((T, G, 3), np.float32), gd], axis=1)
        reg_o, _ = orc.feat_site_problem(forces, feat, div, kbt, 0.0)
        assert rel(np.swapaxes(R[:, :n_feat, :], 1, 2).reshape(-1, n_feat), reg_o) < 5e-6  # (the oracle's features are float32)


@pytest.mark.parametrize("fdt,xdt", [(np.float64, np.float32), (np.float32, np.float64)])
def test_feat_contract_mixed_dtypes(fdt, xdt):
    """aggf_feat_contract with forces and features of different dtypes (featlinearmap.py:361-369 under NumPy promotion)."""
    rng = np.random.default_rng(6)
    T, N, n_feat = 203, 11, 9
    forces = rng.standard_normal((T, N, 3)).astype(fdt)
    feat = rng.standard_normal((T, N, n_feat)).astype(xdt)
    div = rng.standard_normal((T, n_feat, 3)).astype(xdt)
    got = K.feat_contract(dev(forces), dev(feat), dev(div), 0.6, 16)
    want = np.einsum("taf,tad->tfd", feat.astype(np.float64), forces.astype(np.float64)) + 0.6 * div
    assert got.dtype == torch.float64 and got.shape == (T, 16, 3)
    # (NumPy forms `0.6 * div` in float32 before the promotion, the kernel in float64: 1e-8 apart)
    assert rel(got[:, :n_feat].cpu().numpy(), want) < 1e-7 and float(got[:, n_feat:].abs().max()) == 0.0


# ------------------------------------------------------------------ K1 streaming kernel: every instantiation
def _dispatch_cases():
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dispatch_cases.json")
    return json.load(open(path))["cases"]


def _chain_constraints(size, n_groups):
    """`n_groups` chains of `size` consecutive atoms at the front (bond pairs along each chain)."""
    return {frozenset([g * size + j, g * size + j + 1]) for g in range(n_groups) for j in range(size - 1)}


@pytest.mark.parametrize("pair", [("float64", "float64"), ("float32", "float64"), ("float32", "float32")])
def test_gram_streaming_kernel_every_instantiation(pair):
    """One system per instantiation of gram_small_kernel<TIn, TC, NV, KBS, waves, W, C> (panel width x frames per stage
    x register pieces x blocks per wave x dtype pair: what make_plan's thresholds can select; the list comes from
    tools/find_dispatch_cases.py, which reads the library's launch table): `reg_mat.T @ reg_mat` with the column sums
    of `@ con_mat` (qplinear.py:66-71) against the oracle, with enough frames that every workgroup runs several stages
    and the last stage is ragged.  The test also checks that each case still reaches the instantiation it was listed
    for -- a changed threshold shows here, and in tests/test_gpu_zz_coverage.py."""
    from aggforce_amd.constraints import group_layout, groups_csr

    npd = {"float32": np.float32, "float64": np.float64}
    cases = [c for c in _dispatch_cases() if (c["in"], c["compute"]) == pair]
    assert len(cases) >= 40
    rng = np.random.default_rng(7)
    moved = []
    for c in cases:
        N, size, ng = c["N"], c["group_size"], c["n_groups"]
        # frames: > 2 stages for each of the <= 512 workgroups where that stays cheap on the host, ragged end
        T = 2 * 512 * 8 + 37 if N <= 200 else 1037
        forces = (30 * rng.standard_normal((T, N, 3))).astype(npd[c["in"]])
        cons = _chain_constraints(size, ng) if ng else set()
        goa, n_red = group_layout(N, cons)
        assert n_red == c["n_red"]
        gp = ga = None
        if n_red != N:
            p, a = groups_csr(goa, n_red)
            gp, ga = dev(p), dev(a)
        _lib.load().aggf_coverage_reset()
        G = K.gram(dev(forces), gp, ga, n_red, TD[npd[c["compute"]]])
        ran = launched("gram_small_kernel")
        if ran != [c["kernel"]]:
            moved.append((c["kernel"], ran))
        ref = orc.linear_problem(forces, np.eye(N), cons)["qp_mat"]
        tol = 1e-12 if c["compute"] == "float64" else 3e-5
        assert rel(G.cpu().numpy(), ref) < tol and torch.equal(G, G.T), c
    assert not moved, f"cases no longer reach the instantiation they were listed for (re-run tools/find_dispatch_cases.py): {moved[:5]}"
