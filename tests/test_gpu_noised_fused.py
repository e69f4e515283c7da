"""GPU: the noised map (joptgauss_map, qp/jgauss.py:27-140) WITHOUT the extended (T, N + n_cg, 3) arrays.

When N and n_cg are multiples of 128 the fit reads the forces where they lie (aggf_gram_pair on [F | Fa]), transforms
the Gram matrix (aggf_augmented_gram, aggf_sym_group_reduce) and the returned map is applied as
W_N F + (W_a - W_N C') Fa.  Checked against the CPU oracle (which concatenates, like the reference) with injected
noise, and against the general path of the product itself."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from aggforce_amd import LinearMap, Trajectory, joptgauss_map, project_forces  # noqa: E402
from aggforce_amd import _kernels as K  # noqa: E402
from aggforce_amd.qp import gauss as gauss_mod  # noqa: E402
from oracle import aggforce_oracle as orc  # noqa: E402

KBT, VAR = 0.6955215, 0.04


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def case(name):
    rng = np.random.default_rng(31)
    if name == "slice_f32":
        T, N, n_cg, dt = 700, 256, 128, np.float32
        cmat = orc.list_mapping_matrix([[2 * i] for i in range(n_cg)], N)
        cons, l2 = set(), 0.0
    elif name == "dense_f64_constraints":
        T, N, n_cg, dt = 900, 384, 128, np.float64
        cmat = orc.list_mapping_matrix([[3 * i, 3 * i + 1, 3 * i + 2] for i in range(n_cg)], N)
        cons, l2 = {frozenset([6 * i, 6 * i + 1]) for i in range(40)} | {frozenset([6 * i + 1, 6 * i + 4]) for i in range(0, 40, 3)}, 0.5
    else:
        raise KeyError(name)
    coords = (5 * rng.random((T, N, 3))).astype(dt)
    forces = (30 * rng.standard_normal((T, N, 3))).astype(dt)
    eps = [rng.standard_normal((T, n_cg, 3)).astype(np.float32) for _ in range(2)]
    return coords, forces, cmat, cons, l2, eps, dt


@pytest.mark.parametrize("name", ["slice_f32", "dense_f64_constraints"])
def test_noised_map_without_extended_arrays_matches_oracle_and_general_path(name, monkeypatch):
    coords, forces, cmat, cons, l2, eps, dt = case(name)
    N, n_cg = cmat.shape[1], cmat.shape[0]
    cmap = LinearMap(cmat)
    traj = Trajectory(coords=coords, forces=forces)
    calls = {"pair": 0}
    real_pair = K.gram_pair
    monkeypatch.setattr(K, "gram_pair", lambda a, b: (calls.__setitem__("pair", calls["pair"] + 1), real_pair(a, b))[1])
    tm = joptgauss_map(traj, cmap, var=VAR, kbt=KBT, constraints=cons, noise=list(eps), l2_regularization=l2)
    assert calls["pair"] == 1                                           # the fit took the in-place path
    W = tm.tmap.force_map.standard_matrix
    o = orc.joptgauss_force_map(coords, forces, cmat, VAR, KBT, eps[0], cons, l2, dtype=np.float32 if dt == np.float32 else np.float64)
    # float32: MFMA products in float32 against the oracle's float64 Gram (BASELINE's float32 bound); float64
    # trajectories: the augmenter still works in float32 like the reference's JCondNormal (jaxgausstraj.py:202-206),
    # the oracle here in float64 -- the generated sites differ at the 1e-7 level
    tol = 1e-3 if dt == np.float32 else 2e-5
    assert W.shape == (n_cg, N + n_cg) and rel(W, o["force_map"]) < tol, rel(W, o["force_map"])
    assert np.max(np.abs(o["aug_coord_matrix"] @ W.T - np.eye(n_cg))) < 1e-8
    # application: fresh noise eps[1]; the oracle concatenates and applies W to the extended forces
    big = {"n": 0}
    real_aug = K.condnormal_augment
    monkeypatch.setattr(K, "condnormal_augment", lambda *a, **k: (big.__setitem__("n", big["n"] + 1), real_aug(*a, **k))[1])
    mapped = tm(traj)
    assert big["n"] == 0                                                # no extended array was built
    full_c, full_f = orc.augment(coords, forces, cmat, VAR, KBT, eps[1], dtype=np.float32 if dt == np.float32 else np.float64)
    want_f = orc.linearmap_apply(full_f, o["force_map"])
    assert rel(mapped.coords, full_c[:, N:, :]) < 1e-6
    assert rel(mapped.forces, want_f) < tol
    # ... and the product's own general path (extended arrays, aggf_gram on them) with the same noise
    monkeypatch.setattr(gauss_mod, "_joptgauss_without_extended_arrays", lambda *a, **k: None)
    gen = joptgauss_map(traj, cmap, var=VAR, kbt=KBT, constraints=cons, noise=list(eps), l2_regularization=l2)
    assert calls["pair"] == 1
    # (float64 trajectories: the general path adds the source-site correction kbt C' r after rounding it to the
    # augmenter's float32; here it enters through float64 matrix algebra)
    assert rel(W, gen.tmap.force_map.standard_matrix) < (1e-3 if dt == np.float32 else 1e-7)
    monkeypatch.setattr(type(gen), "_call_without_extended_arrays", lambda self, t: None)
    mapped_gen = gen(traj)
    assert big["n"] == 2                                                # the general fit and its application
    assert rel(mapped_gen.forces, mapped.forces) < (1e-3 if dt == np.float32 else 1e-7)
    assert rel(mapped_gen.coords, mapped.coords) < 1e-6


def test_gram_pair_and_augmented_gram_kernels():
    """aggf_gram_pair == aggf_gram of the concatenation -- bit for bit where both run the tile kernel (same tiles, same
    order), to rounding at 256 columns, where aggf_gram takes the 256-column streaming kernel; aggf_augmented_gram and
    aggf_sym_group_reduce against dense NumPy algebra."""
    rng = np.random.default_rng(5)
    for dt, T, N, N2 in [(torch.float64, 333, 256, 128), (torch.float32, 1000, 384, 256), (torch.float64, 64, 128, 128)]:
        a = torch.from_numpy(rng.standard_normal((T, N, 3))).to(dt).cuda()
        b = torch.from_numpy(rng.standard_normal((T, N2, 3))).to(dt).cuda()
        assert K.gram_pair_ok(a, b)
        Gp = K.gram_pair(a, b)
        Gc = K.gram(torch.cat([a, b], dim=1).contiguous(), None, None, N + N2, dt)
        if N + N2 > 256:
            assert torch.equal(Gp, Gc)
        else:
            assert float((Gp - Gc).abs().max()) < 1e-12 * float(Gc.abs().max())
    assert not K.gram_pair_ok(a[:, :100].contiguous(), b)
    n, n2 = 200, 56
    X = rng.standard_normal((n + n2, n + n2))
    Gx = X @ X.T
    C = np.where(rng.random((n2, n)) < 0.03, rng.standard_normal((n2, n)), 0.0)
    cols = K.premap_columns(C, torch.float64, "cuda")
    got = K.augmented_gram(torch.from_numpy(Gx).cuda(), n, cols).cpu().numpy()
    Tm = np.block([[np.eye(n), np.zeros((n, n2))], [-C, np.eye(n2)]])
    assert rel(got, Tm.T @ Gx @ Tm) < 1e-13 and np.array_equal(got, got.T)
    cons = {frozenset([0, 5]), frozenset([5, 9]), frozenset([20, 21, 22]), frozenset([100, 255])}
    Cm = orc.make_bond_constraint_matrix(n + n2, cons)
    from aggforce_amd.constraints import group_layout, groups_csr

    goa, n_red = group_layout(n + n2, cons)
    gp, ga = groups_csr(goa, n_red)
    red = K.sym_group_reduce(torch.from_numpy(Gx).cuda(), torch.from_numpy(gp).cuda(), torch.from_numpy(ga).cuda(), n_red)
    assert rel(red.cpu().numpy(), Cm.T @ Gx @ Cm) < 1e-13


def test_project_forces_with_noised_method_in_place():
    """Through project_forces (method=joptgauss_map) on device arrays, Philox noise: feasibility, finiteness and the
    residual identity mean(mapped^2) ~ x'G x / (3 T n_cg) up to the fresh noise of the application."""
    T, N, n_cg = 20000, 512, 128
    forces = K.synth_normal(T, N, torch.float32, 1, sigma=30.0)
    coords = K.synth_normal(T, N, torch.float32, 2, sigma=0.3, lattice=1.5)
    cmap = LinearMap([[4 * i] for i in range(n_cg)], n_fg_sites=N)
    out = project_forces(coords, forces, cmap, constrained_inds=None, method=joptgauss_map, var=0.01, kbt=KBT, seed=3)
    W = out["tmap"].tmap.force_map.standard_matrix
    assert W.shape == (n_cg, N + n_cg) and np.max(np.abs(W[:, N:] - np.eye(n_cg))) < 1e-9
    assert tuple(out["mapped_forces"].shape) == (T, n_cg, 3) and bool(torch.isfinite(out["mapped_forces"]).all())
    assert tuple(out["mapped_coords"].shape) == (T, n_cg, 3)
    dev = out["mapped_coords"].double() - coords[:, ::4, :][:, :n_cg].double()
    assert abs(dev.var().item() / 0.01 - 1.0) < 2e-2
    assert np.isfinite(out["residual"]) and out["residual"] > 0


def test_generated_sites_equal_the_extended_arrays_bit_for_bit():
    """aggf_condnormal_sites draws the noise aggf_condnormal_augment draws (same Philox stream and expressions), also
    for shards whose first element is not quad-aligned in the global stream."""
    from aggforce_amd.trajectory import CondNormal

    rng = np.random.default_rng(2)
    for T, N, n_cg, off in [(257, 40, 5, 3), (100, 64, 16, 0), (33, 9, 3, 1_000_001)]:
        coords = rng.random((T, N, 3)).astype(np.float32)
        forces = rng.standard_normal((T, N, 3)).astype(np.float32)
        cmap = LinearMap([[i, i + 1] for i in range(n_cg)], n_fg_sites=N)
        a = CondNormal(var=0.09, premap=cmap, seed=77, frame_offset=off)
        b = CondNormal(var=0.09, premap=cmap, seed=77, frame_offset=off)
        for _ in range(2):                                  # two draws: the per-call stream offset advances alike
            oc, of = a.augment_trajectory(coords, forces, KBT)
            y, fa, _ = b.noise_sites(coords, KBT)
            assert np.array_equal(oc[:, N:, :], y.cpu().numpy()) and np.array_equal(of[:, N:, :], fa.cpu().numpy())


@pytest.mark.parametrize("name", ["slice_f32", "dense_f64_constraints"])
def test_noised_grid_cv_one_pass(name):
    """project_forces_grid_cv over l2_regularization of joptgauss_map in one pass (one noise realisation, per-fold
    Gram matrices of the extended system read in place): (1) with the realisation fixed, equal to the LINEAR one-pass
    cross-validation -- itself checked against the loop and the oracle in test_gpu_staged.py -- run on the extended
    trajectory built with the same noise; (2) against the reference's loop, which draws fresh noise for every fit and
    application, statistically."""
    from aggforce_amd import agg
    from aggforce_amd.map import lmap_augvariables
    from aggforce_amd.trajectory import AugmentedTrajectory, CondNormal

    coords, forces, cmat, cons, _, eps, dt = case(name)
    cmap = LinearMap(cmat)
    grid = {"l2_regularization": [0.0, 1.0, 50.0]}
    calls = {"n": 0}
    real = agg._score_folds

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    agg._score_folds = counted
    try:
        fast = agg.project_forces_grid_cv(grid, coords, forces, n_folds=4, rng=np.random.default_rng(2), cv_noise=eps[0],
                                          coord_map=cmap, constrained_inds=cons, method=joptgauss_map, var=VAR, kbt=KBT)
    finally:
        agg._score_folds = real
    assert calls["n"] == 1  # the one-pass form ran
    aug = AugmentedTrajectory.from_trajectory(t=Trajectory(coords=coords, forces=forces), kbt=KBT,
                                              augmenter=CondNormal(var=VAR, premap=cmap).inject_noise(eps[0]))
    lin = agg.project_forces_grid_cv(grid, aug.coords, aug.forces, n_folds=4, rng=np.random.default_rng(2),
                                     coord_map=lmap_augvariables(aug), constrained_inds=cons)
    tol = 2e-4 if dt == np.float32 else 1e-6  # float32: products in float32 on differently rounded operands
    for key in lin["scores"]:
        assert fast["n_runs"][key] == lin["n_runs"][key] == 4
        assert abs(fast["scores"][key] - lin["scores"][key]) < tol * abs(lin["scores"][key]), key
    loop = agg.project_forces_grid_cv(grid, coords, forces, n_folds=4, rng=np.random.default_rng(2), reuse_gram=False,
                                      coord_map=cmap, constrained_inds=cons, method=joptgauss_map, var=VAR, kbt=KBT, seed=5)
    for key in loop["scores"]:
        assert loop["n_runs"][key] == 4
        assert abs(fast["scores"][key] - loop["scores"][key]) < 0.05 * abs(loop["scores"][key]), key


def _noised_cv_rank_worker(rank, world, port, out_dir):
    """Two processes on one GPU (gloo): each rank holds half of the frames, splits them into folds locally and draws
    the noise of ITS frames; global fold k is the union of the ranks' fold k."""
    import os
    import sys

    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from aggforce_amd import LinearMap as LM
    from aggforce_amd import joptgauss_map as jm
    from aggforce_amd.agg import project_forces_grid_cv as cv

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    coords, forces, cmat, cons, _, eps, _ = case("dense_f64_constraints")
    half = slice(rank * 450, (rank + 1) * 450)
    res = cv({"l2_regularization": [0.0, 2.0]}, coords[half], forces[half], n_folds=3, rng=np.random.default_rng(50 + rank),
             cv_noise=eps[0][half], coord_map=LM(cmat), constrained_inds=cons, method=jm, var=VAR, kbt=KBT, comm=True)
    out = [(res["scores"][k], res["sds"][k], res["n_runs"][k]) for k in res["scores"]]
    np.save(os.path.join(out_dir, f"ncv{rank}.npy"), np.array(out, dtype=np.float64))
    dist.destroy_process_group()


def test_noised_grid_cv_two_ranks_match_one_process_on_union_folds(tmp_path):
    import socket

    import torch.multiprocessing as mp

    from aggforce_amd import agg
    from aggforce_amd.agg import mean, process_cvargs, sample_sd
    from aggforce_amd.qp.gauss import cv_joptgauss_fold_grams

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_noised_cv_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "ncv0.npy"), np.load(tmp_path / "ncv1.npy")
    assert np.array_equal(r0, r1)  # both ranks hold the same all-reduced matrices and solve the same problems
    coords, forces, cmat, cons, _, eps, _ = case("dense_f64_constraints")
    local = []
    for rank in range(2):
        fr = np.arange(450)
        np.random.default_rng(50 + rank).shuffle(fr)
        local.append([f + 450 * rank for f in np.array_split(fr, 3)])
    folds = [np.concatenate([local[0][k], local[1][k]]) for k in range(3)]
    grams, prob = cv_joptgauss_fold_grams(coords, forces, LinearMap(cmat), VAR, KBT, cons, None, folds, noise=eps[0])
    grid = process_cvargs({"l2_regularization": [0.0, 2.0]})
    one = agg._score_folds(grid, grams, [float(len(f)) for f in folds], prob, {})
    for row, (label, _) in zip(r0, grid):
        assert row[2] == one["n_runs"][label] == 3
        assert abs(row[0] - one["scores"][label]) < 1e-7 * abs(one["scores"][label])
        assert abs(row[1] - one["sds"][label]) < 1e-5 * abs(one["sds"][label])
    assert mean([1.0, 3.0]) == 2.0 and sample_sd([1.0, 3.0]) > 0  # (the helpers the rows were built with)


def _mixed_layout_rank_worker(rank, world, port, out_dir, mode):
    """rank 1's shard is a strided device view (``noncontig``) or has no frames (``empty``); rank 0's is
    plain.  The fused / general choice must be taken jointly: the two paths all-reduce different matrices."""
    import os
    import sys

    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from aggforce_amd import LinearMap as LM
    from aggforce_amd import Trajectory as TJ
    from aggforce_amd import joptgauss_map as jm

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    coords, forces, cmat, cons, l2, eps, _ = case("dense_f64_constraints")
    cut = 900 if mode == "empty" else 450
    sl = slice(0, cut) if rank == 0 else slice(cut, 900)
    c, f = torch.from_numpy(coords[sl]).cuda(), torch.from_numpy(forces[sl]).cuda()
    if mode == "noncontig" and rank == 1:
        wide = torch.zeros((f.shape[0], f.shape[1], 4), dtype=f.dtype, device=f.device)
        wide[:, :, :3] = f
        f = wide[:, :, :3]  # same values, strides (4 N, 4, 1): not a layout aggf_gram_pair reads in place
        assert not f.is_contiguous()
    tm = jm(TJ(coords=c, forces=f), LM(cmat), var=VAR, kbt=KBT, constraints=cons, noise=[eps[0][sl]],
            l2_regularization=l2, frame_offset=sl.start, comm=True)
    np.save(os.path.join(out_dir, f"mixed_{mode}_{rank}.npy"), tm.tmap.force_map.standard_matrix)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["noncontig", "empty"])
def test_fused_noised_fit_two_ranks_agree_on_the_path(tmp_path, mode):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_mixed_layout_rank_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    w0, w1 = np.load(tmp_path / f"mixed_{mode}_0.npy"), np.load(tmp_path / f"mixed_{mode}_1.npy")
    assert np.array_equal(w0, w1)  # replicated solve of one all-reduced matrix
    coords, forces, cmat, cons, l2, eps, _ = case("dense_f64_constraints")
    o = orc.joptgauss_force_map(coords, forces, cmat, VAR, KBT, eps[0], cons, l2, dtype=np.float64)
    assert rel(w0, o["force_map"]) < 2e-5
