"""GPU: the per-frame contraction kernels (K3c trjdot with a 3-D factor, K4b constraint rows, K4c dense
featuriser contractions) through the C ABI against the CPU oracle, the generic (non-fused) featurised fit
with a user-written featuriser, and the featurised fit with frames sharded over two ranks."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from aggforce_amd import LinearMap, Trajectory  # noqa: E402
from aggforce_amd import _kernels as K  # noqa: E402
from aggforce_amd.map import CLAMap  # noqa: E402
from aggforce_amd.qp import Multifeaturize, gb_feat, id_feat, qp_feat_linear_map  # noqa: E402
from aggforce_amd.util import Curry, trjdot  # noqa: E402
from oracle import aggforce_oracle as orc  # noqa: E402

KBT = 0.6955215


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("pdt,fdt", [(np.float32, np.float32), (np.float64, np.float32), (np.float32, np.float64),
                                     (np.float64, np.float64)])
def test_trjdot_per_frame_factor_matches_oracle(pdt, fdt):
    """util.trjdot's "...fd,...cf->...cd" branch (util.py:119-125): numpy promotion, awkward sizes."""
    rng = np.random.default_rng(5)
    for T, N, n_cg in [(1, 1, 1), (7, 13, 5), (33, 64, 4), (20, 131, 9), (5, 1000, 2), (64, 257, 66)]:
        points = (10 * rng.standard_normal((T, N, 3))).astype(pdt)
        factor = rng.standard_normal((T, n_cg, N)).astype(fdt)
        ref = orc.trjdot(points.astype(np.float64), factor.astype(np.float64))
        out = trjdot(points, factor)
        assert isinstance(out, np.ndarray) and out.shape == (T, n_cg, 3)
        assert out.dtype == np.result_type(pdt, fdt)
        tol = 1e-12 if out.dtype == np.float64 and pdt == fdt == np.float64 else 3e-6
        assert rel(out, ref) < tol, (T, N, n_cg, rel(out, ref))
        on_dev = trjdot(dev(points), dev(factor))
        assert on_dev.is_cuda and np.array_equal(on_dev.cpu().numpy(), out)
    with pytest.raises(ValueError):
        trjdot(points, factor[0, 0])  # 1-D factor
    with pytest.raises(ValueError):
        trjdot(points, factor[:, :, :-1])  # site count mismatch


def test_clamap_call_with_user_scale_and_trans_matches_reference_formula():
    """CLAMap.__call__ = trjdot(points, scale(copoints)) + trans(copoints) (map/core.py:428-430)."""
    rng = np.random.default_rng(6)
    T, N, n_cg = 41, 37, 6
    points = rng.standard_normal((T, N, 3))
    copoints = rng.standard_normal((T, N, 3))
    mix = rng.standard_normal((n_cg, N))

    def scale(y):  # (T, n_cg, N): configuration-dependent weights
        return mix[None] * (1.0 + 0.1 * np.tanh(np.asarray(y)[:, None, :, 0]))

    def trans(y):
        return np.asarray(y)[:, :n_cg, :] * 0.25

    cla = CLAMap(scale=scale, trans=trans, n_fg_sites=N)
    assert cla.n_cg_sites == n_cg
    out = cla(points, copoints)
    ref = orc.trjdot(points, scale(copoints)) + trans(copoints)
    assert out.shape == (T, n_cg, 3) and rel(out, ref) < 1e-13
    out32 = cla(points.astype(np.float32), copoints)  # float32 points, float64 map: promoted like NumPy
    assert out32.dtype == np.float64 and rel(out32, orc.trjdot(points.astype(np.float32), scale(copoints)) + trans(copoints)) < 1e-12
    # the reference's expression also takes a frame-independent (2-D) scale and a broadcastable trans
    # (a map that passes the constructor's zero test must not fail when called)
    for tr in (lambda y: 0.0, lambda y: np.full((n_cg, 3), 0.5), lambda y: np.asarray(y)[:, :n_cg, :] * 0.25):
        flat = CLAMap(scale=lambda y: mix, trans=tr, n_fg_sites=N)
        assert flat.n_cg_sites == n_cg
        want = orc.trjdot(points, mix) + tr(copoints)
        got = flat(points, copoints)
        assert got.shape == (T, n_cg, 3) and rel(got, want) < 1e-13
    per_frame_bcast = CLAMap(scale=scale, trans=lambda y: np.full((1, n_cg, 3), -1.5), n_fg_sites=N)
    assert rel(per_frame_bcast(points, copoints), orc.trjdot(points, scale(copoints)) - 1.5) < 1e-13


def _dense_featuriser(points, cmap, constraints):
    """A user-written featuriser following the reference protocol (featlinearmap.py:49-67): per site, a
    one-hot atom "type" (a % 7; makes the sampled constraint rows feasible, as id_feat does for the
    reference) and two smooth functions of the distance to the mapped site, with their divergences."""
    pts = np.asarray(points, dtype=np.float64)
    T, N, _ = pts.shape
    cg = orc.linearmap_apply(pts, np.asarray(cmap.standard_matrix, dtype=np.float64))
    types = np.zeros((T, N, 7))
    types[:, np.arange(N), np.arange(N) % 7] = 1.0
    feats, divs = [], []
    for c in range(cmap.n_cg_sites):
        disp = pts - cg[:, c:c + 1, :]
        r2 = (disp ** 2).sum(-1)
        f = np.concatenate([types, np.exp(-r2 / 9.0)[..., None], (1.0 / (1.0 + r2))[..., None]], axis=-1)
        d1 = (np.exp(-r2 / 9.0) * (-2.0 / 9.0))[..., None] * disp
        d2 = (-2.0 / (1.0 + r2) ** 2)[..., None] * disp
        d = np.concatenate([np.zeros((T, 7, 3)), d1.sum(1)[:, None, :], d2.sum(1)[:, None, :]], axis=1)
        feats.append(f.astype(np.float32))
        divs.append(d.astype(np.float32))  # (T, n_feat, 3)
    return {"feats": feats, "divs": divs, "names": [f"type{i}" for i in range(7)] + ["gauss", "lorentz"]}


def _feat_system(T=90, N=21, seed=1, dtype=np.float32):
    rng = np.random.default_rng(seed)
    coords = (5 * rng.random((T, N, 3)) + 1).astype(dtype)
    forces = (20 * rng.standard_normal((T, N, 3))).astype(dtype)
    cmat = orc.list_mapping_matrix([[0, 1], [5, 6, 7], [12], [19, 20]], N)
    return coords, forces, cmat


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dense_featuriser_kernels_match_oracle(dtype):
    coords, forces, cmat = _feat_system(dtype=dtype)
    cmap = LinearMap(cmat)
    res = _dense_featuriser(coords, cmap, set())
    feat, div = res["feats"][1], res["divs"][1]
    T, N, n_feat = feat.shape
    # K4c regression matrix (featlinearmap.py:361-369), with and without K1 padding
    reg_o, qp_o = orc.feat_site_problem(forces.astype(np.float64), feat.astype(np.float64), div.astype(np.float64), KBT, 0.0)
    for ld in (None, 128):
        r3 = K.feat_contract(dev(forces), dev(feat), dev(div), KBT, ld)
        got = np.swapaxes(r3.cpu().numpy()[:, :n_feat, :], 1, 2).reshape(-1, n_feat)
        assert r3.dtype == (torch.float64 if dtype == np.float64 else torch.float32)
        assert rel(got, reg_o) < (1e-13 if dtype == np.float64 else 2e-6)
        if ld:
            assert r3.shape == (T, 128, 3) and float(r3[:, n_feat:, :].abs().max()) == 0.0
            assert rel(K.gram(r3, None, None, n_feat, torch.float64).cpu().numpy(), qp_o) < 1e-5
    # K4b constraint rows (featlinearmap.py:445-459)
    idx = np.array([3, 77, 0, 41, 89])
    for site in (0, 3):
        A, b = K.feat_constraint_rows(dev(feat), idx, dev(cmat), site)
        Ao, bo = orc.feat_constraint_arrays(feat.astype(np.float64), site, cmat, idx)
        assert A.shape == (5 * 4, n_feat) and rel(A.cpu().numpy(), Ao) < 1e-13
        assert np.array_equal(b.cpu().numpy().ravel(), bo)
    with pytest.raises(IndexError):
        K.feat_constraint_rows(dev(feat), np.array([T]), dev(cmat), 0)
    # scale_f weights (featlinearmap.py:512-515)
    coef = np.random.default_rng(2).standard_normal(n_feat)
    w = torch.zeros((T, 4, N), dtype=torch.float64, device="cuda")
    K.feat_weights(dev(feat), dev(coef), w, 2)
    assert rel(w[:, 2, :].cpu().numpy(), np.einsum("...ij,j->...i", feat.astype(np.float64), coef)) < 1e-13
    assert float(w[:, [0, 1, 3], :].abs().max()) == 0.0


def test_gb_constraint_rows_match_dense_oracle():
    """K4b for the fused [id | gb] features against _constr_arrays on the dense one-hot tensor."""
    rng = np.random.default_rng(8)
    T, N = 30, 14
    coords = (6 * rng.random((T, N, 3)) + 1).astype(np.float32)
    cons = {frozenset([1, 2]), frozenset([4, 5]), frozenset([5, 6]), frozenset([10, 13])}
    cmat = orc.list_mapping_matrix([[0, 1], [4, 7], [8, 9], [12, 13]], N)
    cmap = LinearMap(cmat)
    ids = orc.id_feat_ids(N, cons)
    G = int(ids.max()) + 1
    kw = dict(outer=8.0, inner=0.0, n_basis=4, width=1.0, dist_power=0.5)
    smear = orc.smear_matrix(orc.reduce_constraint_sets(cons), N)
    cg = orc.linearmap_apply(coords, cmat)
    onehot = np.zeros((T, N, G), dtype=np.float32)
    onehot[:, np.arange(N), ids] = 1
    from aggforce_amd.qp.gbfeat import CLIP, _Geometry, gb_centers

    geo = _Geometry(coords, cmap, cons, True)
    assert np.array_equal(geo.ids, ids)
    Mg = torch.from_numpy(np.ascontiguousarray(geo.Mg)).cuda()
    centers = torch.from_numpy(gb_centers(8.0, 0.0, 4, 0.5)).cuda()
    idx = np.array([2, 29, 11])
    sel = torch.as_tensor(idx, device="cuda")
    for site in range(4):
        gf, _ = orc.gb_feat_site(coords, cg[:, site, :], ids, smear, n_channels=G - 1, **kw)
        dense = np.concatenate([onehot, gf], axis=2)
        Ao, bo = orc.feat_constraint_arrays(dense.astype(np.float64), site, cmat, idx)
        gauss, _ = K.gb_channels(geo.Pg[sel].contiguous(), geo.cg[sel].contiguous(), site, geo.sizes, G - 1, centers, 1.0, CLIP)
        A, b = K.gb_constraint_rows(Mg, gauss, 3, G, G - 1, 4, site)
        assert A.shape == Ao.shape and rel(A.cpu().numpy(), Ao) < 2e-6
        assert np.array_equal(b.cpu().numpy().ravel(), bo)
    A_id, _ = K.gb_constraint_rows(Mg, None, 3, G, 0, 1, 0)  # id features only
    assert rel(A_id.cpu().numpy(), orc.feat_constraint_arrays(onehot.astype(np.float64), 0, cmat, idx)[0]) < 1e-15


@pytest.mark.parametrize("gdt", [torch.float32, torch.float64])
def test_gb_constraint_gram_matches_the_product_of_the_rows(gdt):
    """aggf_gb_constraint_gram forms A'A from the structure of the rows (S multiply-adds per entry): against the
    float64 product of the rows aggf_gb_constraint_rows writes -- all columns and a compacted selection, more
    sampled frames than one LDS chunk, a padded leading dimension; then the batched solve with that A'A against
    the batched solve forming it itself."""
    rng = np.random.default_rng(31)
    n_cg, G, nb, S = 7, 150, 5, 45
    n_ch = G - 1
    Mg = torch.from_numpy(rng.random((n_cg, G)) * (rng.random((n_cg, G)) < 0.3)).cuda()
    gauss = torch.from_numpy(rng.random((S, n_ch, nb))).to(gdt).cuda()
    M2 = K.gb_group_overlap(Mg)
    assert rel(M2.cpu().numpy(), (Mg.T @ Mg).cpu().numpy()) < 1e-14
    keep = np.sort(rng.choice(n_ch * nb, size=300, replace=False)).astype(np.int32)
    for cols, n_id in ((None, G), (torch.from_numpy(keep).cuda(), G), (torch.from_numpy(keep).cuda(), 0)):
        A, _ = K.gb_constraint_rows(Mg, gauss, S, n_id, n_ch, nb, 2, cols=cols)
        n = A.shape[1]
        ld = n + 37
        out = torch.full((ld, ld), float("nan"), dtype=torch.float64, device="cuda")
        K.gb_constraint_gram(M2, gauss, S, n_id, n_ch, nb, out, cols=cols)
        got = out.cpu().numpy()
        want = np.zeros((ld, ld))
        want[:n, :n] = (A.T @ A).cpu().numpy()
        lower = np.tril(np.ones((ld, ld), dtype=bool))
        assert np.isfinite(got[lower]).all()
        assert np.max(np.abs(got[lower] - want[lower])) < 1e-13 * np.max(np.abs(want))
    # the solve: same minimiser whichever way the shift was formed
    p, n, m = 3, 90, 24
    R = rng.standard_normal((p, 200, n))
    Gs = torch.from_numpy(np.einsum("ptn,ptm->pnm", R, R)).cuda()
    As = torch.from_numpy(rng.standard_normal((p, m, n))).cuda()
    bs = torch.from_numpy(rng.standard_normal((p, m, 1))).cuda()
    AtA = torch.tril(As.transpose(1, 2) @ As).contiguous()  # the lower triangle is all that is read
    X0, st0 = K.eq_qp_solve_batched(Gs, 0.5, None, As, bs, schur_reg=1e-12, n_refine=2)
    X1, st1 = K.eq_qp_solve_batched(Gs, 0.5, None, As, bs, schur_reg=1e-12, n_refine=2, AtA=AtA)
    assert float(st1[:, 0].abs().max()) == 0.0 and float(st1[:, 1].max()) < 1e-10
    assert rel(X1.cpu().numpy(), X0.cpu().numpy()) < 1e-10
    with pytest.raises(ValueError):
        K.eq_qp_solve_batched(Gs, 0.5, None, As, bs, AtA=AtA[:, :-1])
    # sparse constraint rows: the variables they touch taken last, the forward solve restricted to the rows below
    p, n, m, nt = 3, 700, 40, 150
    R = rng.standard_normal((p, 900, n))
    Gs = torch.from_numpy(np.einsum("ptn,ptm->pnm", R, R)).cuda()
    A_h = np.zeros((p, m, n))
    perm_h = np.empty((p, n), dtype=np.int32)
    for q in range(p):
        t = np.sort(rng.choice(n, size=nt - 7 * q, replace=False))  # a different number per problem
        A_h[q][:, t] = rng.standard_normal((m, len(t)))
        mask = np.zeros(n, dtype=bool)
        mask[t] = True
        perm_h[q] = np.concatenate([np.nonzero(~mask)[0], np.nonzero(mask)[0]])
    As = torch.from_numpy(A_h).cuda()
    bs = torch.from_numpy(rng.standard_normal((p, m, 2))).cuda()
    AtA = torch.tril(As.transpose(1, 2) @ As).contiguous()
    X0, st0 = K.eq_qp_solve_batched(Gs, 0.3, None, As, bs, schur_reg=1e-12, n_refine=2, AtA=AtA)
    perm = torch.from_numpy(perm_h).cuda()
    for first in (0, n - nt, 300):  # 0: permutation only; n - nt: blocks 0..1 skipped; 300: block 0 skipped
        X1, st1 = K.eq_qp_solve_batched(Gs, 0.3, None, As, bs, schur_reg=1e-12, n_refine=2, AtA=AtA, perm=perm,
                                        a_first_col=first)
        assert float(st1[:, 0].abs().max()) == 0.0 and float(st1[:, 1].max()) < 1e-10
        assert rel(X1.cpu().numpy(), X0.cpu().numpy()) < 1e-9, first
    with pytest.raises(ValueError):
        K.eq_qp_solve_batched(Gs, 0.3, None, As, bs, perm=perm)  # needs AtA
    with pytest.raises(ValueError):
        K.eq_qp_solve_batched(Gs, 0.3, None, As, bs, AtA=AtA, perm=perm[:, :-1].contiguous())


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_generic_featuriser_fit_and_apply_match_oracle(dtype):
    """qp_feat_linear_map with a featuriser the library has never seen: every contraction is a HIP kernel
    (K4b rows, K4c regression matrix, K1, K2, K4c/K3 application); compared with the oracle end to end."""
    coords, forces, cmat = _feat_system(T=120, dtype=dtype)
    cmap = LinearMap(cmat)
    traj = Trajectory(coords=coords, forces=forces)
    rng = np.random.default_rng(4)
    frames = [rng.choice(120, size=1, replace=False) for _ in range(4)]  # one frame: well-conditioned rows
    tm = qp_feat_linear_map(traj, cmap, _dense_featuriser, KBT, constraints=set(), frame_indices=frames,
                            l2_regularization=10.0)
    res = _dense_featuriser(coords, cmap, set())
    f64 = forces.astype(np.float64)
    ocoef = orc.qp_feat_linear_map(f64, cmat, [f.astype(np.float64) for f in res["feats"]],
                                   [d.astype(np.float64) for d in res["divs"]], KBT, frames, 10.0)
    coefs = np.stack(tm.force_map.tags["coef_list"])
    assert tm.force_map.tags["feat_names"][-2:] == ["gauss", "lorentz"]
    # float32 inputs: the regression matrix is float32 as in the reference (featlinearmap.py:361-369 on
    # float32 arrays); BASELINE's float32 tolerance is 1e-3.  float64: the same arithmetic as the oracle.
    tol = 1e-3 if dtype == np.float32 else 1e-6
    assert rel(coefs, np.stack(ocoef)) < tol
    mapped = tm(traj)
    oref = orc.cla_apply(f64, [f.astype(np.float64) for f in res["feats"]], [d.astype(np.float64) for d in res["divs"]], ocoef)
    assert rel(mapped.forces, oref) < tol
    # scale/trans (each re-runs the featuriser, as in the reference) reproduce the fused application
    fm = tm.force_map
    by_parts = orc.trjdot(f64, fm.scale(coords)) + fm.trans(coords)
    assert rel(by_parts, mapped.forces) < 1e-6
    # constraint rows hold on the sampled frames
    sc = fm.scale(coords[frames[2]])
    assert np.max(np.abs(np.einsum("tca,da->tcd", sc, cmat)[:, 2, :] - np.eye(4)[2])) < 1e-7


# ------------------------------------------------------------------ featurised fit, frames sharded over 2 ranks
def _feat_rank_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from aggforce_amd import LinearMap as LM, Trajectory as Tr
    from aggforce_amd import _kernels as KK
    from aggforce_amd.distributed import frame_shard
    from aggforce_amd.qp import Multifeaturize as MF, gb_feat as gb, id_feat as idf, qp_feat_linear_map as fit
    from aggforce_amd.util import Curry as Cu

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    T, N, n_cg = 400, 48, 4
    b, e = frame_shard(T, rank, world)
    forces = KK.synth_normal(e - b, N, torch.float32, 17, frame_offset=b, sigma=30.0)
    coords = KK.synth_normal(e - b, N, torch.float32, 18, frame_offset=b, sigma=0.3, lattice=1.5)
    cmap = LM([[3 * i * 4] for i in range(n_cg)], n_fg_sites=N)
    cons = {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
    feat = MF([idf, Cu(gb, outer=8.0, n_basis=3)])
    # unseeded on purpose, and a different generator state per rank: rank 0's draw must win
    tm = fit(Tr(coords=coords, forces=forces), cmap, feat, 0.6955215, constraints=cons, n_constraint_frames=6,
             rng=np.random.default_rng(100 + rank), comm=True)
    np.save(os.path.join(out_dir, f"coef{rank}.npy"), np.stack(tm.force_map.tags["coef_list"]))
    np.save(os.path.join(out_dir, f"frames{rank}.npy"), np.stack(tm.force_map.tags["constraint_frames"]))
    tg = fit(Tr(coords=coords, forces=forces), cmap, feat, 0.6955215, constraints=cons, n_constraint_frames=6,
             rng=np.random.default_rng(100 + rank), comm=True, fused=False)
    np.save(os.path.join(out_dir, f"gcoef{rank}.npy"), np.stack(tg.force_map.tags["coef_list"]))
    # the ranks estimate the batch size (sites fitted side by side) from their own memory state; the
    # estimates may differ, the batch size in use may not (it fixes the shapes and the count of the all-reduces)
    from aggforce_amd.qp import gbfeat as gbmod

    real = gbmod._sites_per_batch
    gbmod._sites_per_batch = lambda n_cg, n_feat, m, device: (1 if rank == 0 else 3)
    try:
        tb = fit(Tr(coords=coords, forces=forces), cmap, feat, 0.6955215, constraints=cons, n_constraint_frames=6,
                 rng=np.random.default_rng(100 + rank), comm=True)
    finally:
        gbmod._sites_per_batch = real
    np.save(os.path.join(out_dir, f"bcoef{rank}.npy"), np.stack(tb.force_map.tags["coef_list"]))
    np.save(os.path.join(out_dir, f"bbatch{rank}.npy"), np.array([tb.force_map.tags["fit_info"]["sites_per_batch"]]))
    dist.destroy_process_group()


def test_featurised_fit_is_replicated_across_two_ranks(tmp_path):
    """with comm= every rank must solve the SAME problem (same Gram AND same constraint rows)."""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_feat_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    c0, c1 = np.load(tmp_path / "coef0.npy"), np.load(tmp_path / "coef1.npy")
    f0, f1 = np.load(tmp_path / "frames0.npy"), np.load(tmp_path / "frames1.npy")
    assert np.array_equal(f0, f1) and f0.max() >= 200  # global frame numbers, some owned by rank 1
    assert np.array_equal(c0, c1)
    g0, g1 = np.load(tmp_path / "gcoef0.npy"), np.load(tmp_path / "gcoef1.npy")
    assert np.array_equal(g0, g1)
    # per-rank batch-size estimates 1 and 3: both ranks used the minimum and got the same coefficients as before
    assert np.load(tmp_path / "bbatch0.npy")[0] == np.load(tmp_path / "bbatch1.npy")[0] == 1
    b0, b1 = np.load(tmp_path / "bcoef0.npy"), np.load(tmp_path / "bcoef1.npy")
    assert np.array_equal(b0, b1) and rel(b0, c0) < 1e-9
    # single process on the whole trajectory with the same constraint frames
    T, N, n_cg = 400, 48, 4
    forces = K.synth_normal(T, N, torch.float32, 17, sigma=30.0)
    coords = K.synth_normal(T, N, torch.float32, 18, sigma=0.3, lattice=1.5)
    cmap = LinearMap([[3 * i * 4] for i in range(n_cg)], n_fg_sites=N)
    cons = {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
    feat = Multifeaturize([id_feat, Curry(gb_feat, outer=8.0, n_basis=3)])
    one = qp_feat_linear_map(Trajectory(coords=coords, forces=forces), cmap, feat, KBT, constraints=cons,
                             frame_indices=list(f0))
    assert rel(c0, np.stack(one.force_map.tags["coef_list"])) < 1e-6
    assert rel(g0, c0) < 2e-3


# ------------------------------------------------------------------ constrained_inds="auto" with frames sharded over ranks
def _auto_rank_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from aggforce_amd import LinearMap as LM, guess_pairwise_constraints as guess, project_forces as pf
    from aggforce_amd.distributed import frame_shard

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    coords, forces = np.load(os.path.join(out_dir, "coords.npy")), np.load(os.path.join(out_dir, "forces.npy"))
    b, e = frame_shard(coords.shape[0], rank, world)
    cons = guess(coords[b:e], threshold=1e-3, comm=True)
    np.save(os.path.join(out_dir, f"cons{rank}.npy"), np.array(sorted(sorted(c) for c in cons)))
    res = pf(coords[b:e], forces[b:e], LM([[0], [40], [90]], n_fg_sites=coords.shape[1]), comm=True)  # "auto"
    np.save(os.path.join(out_dir, f"W{rank}.npy"), res["tmap"].force_map.standard_matrix)
    dist.destroy_process_group()


def test_auto_constraints_with_sharded_frames(tmp_path):
    """constrained_inds="auto" + comm=: the per-rank pair-distance statistics are pooled exactly (Chan), so every
    rank guesses the set the whole trajectory gives -- including a pair that is rigid on each shard but sits at
    DIFFERENT distances on the two shards (only the pooled variance sees that it moved)."""
    import torch.multiprocessing as mp
    from aggforce_amd import guess_pairwise_constraints, project_forces

    rng = np.random.default_rng(9)
    T, N = 301, 131
    xyz = 10 * rng.random((1, N, 3)) + 0.3 * rng.standard_normal((T, N, 3))
    for a, b_ in ((0, 1), (5, 70), (129, 130)):
        xyz[:, b_, :] = xyz[:, a, :] + rng.standard_normal(3)          # rigid over the whole trajectory
    half = T - T // 2                                                    # frame_shard(301, 0, 2) = [0, 151)
    xyz[:half, 21, :] = xyz[:half, 20, :] + np.array([1.0, 0.0, 0.0])   # rigid on shard 0 at distance 1 ...
    xyz[half:, 21, :] = xyz[half:, 20, :] + np.array([1.5, 0.0, 0.0])   # ... and on shard 1 at distance 1.5
    forces = 20 * rng.standard_normal((T, N, 3))
    np.save(tmp_path / "coords.npy", xyz)
    np.save(tmp_path / "forces.npy", forces)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_auto_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    whole = guess_pairwise_constraints(xyz, threshold=1e-3)
    assert whole == {frozenset([0, 1]), frozenset([5, 70]), frozenset([129, 130])} == orc.guess_pairwise_constraints(xyz, threshold=1e-3)
    want = np.array(sorted(sorted(c) for c in whole))
    c0, c1 = np.load(tmp_path / "cons0.npy"), np.load(tmp_path / "cons1.npy")
    assert np.array_equal(c0, want) and np.array_equal(c1, want)
    # each shard alone WOULD have taken (20, 21) for a constraint
    assert frozenset([20, 21]) in guess_pairwise_constraints(xyz[:half], threshold=1e-3)
    one = project_forces(xyz, forces, LinearMap([[0], [40], [90]], n_fg_sites=N))
    W0, W1 = np.load(tmp_path / "W0.npy"), np.load(tmp_path / "W1.npy")
    assert np.array_equal(W0, W1) and rel(W0, one["tmap"].force_map.standard_matrix) < 1e-9
