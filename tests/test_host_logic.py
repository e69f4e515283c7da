"""CPU: host-side logic of the product (no compute kernels run here), the C ABI export check,
and the frame-sharded all-reduce path under gloo with world_size 2."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

import aggforce_amd
from aggforce_amd import LinearMap
from aggforce_amd.constraints import group_layout, groups_csr, reduce_constraint_sets, constraint_lookup_dict
from aggforce_amd.qp import make_bond_constraint_matrix, constraint_aware_uni_map, id_feat, FeatZipper, Multifeaturize
from aggforce_amd.qp.featlinearmap import constraint_group_labels
from aggforce_amd.map import smear_map
from aggforce_amd.util import Curry, curry, flatten
from aggforce_amd import _lib
from oracle import aggforce_oracle as orc
from conftest import ROOT, cons_from_array, cons_in_insertion_order

CONS_CASES = [
    set(),
    {frozenset([0, 1])},
    {frozenset([0, 1]), frozenset([1, 2]), frozenset([2, 3]), frozenset([18, 20]), frozenset([20, 23])},
    {frozenset([4, 5]), frozenset([5, 9]), frozenset([9, 4]), frozenset([10, 11, 14]), frozenset([14, 15])},
    {frozenset([7, 3]), frozenset([22, 3]), frozenset([8, 21])},
]


@pytest.mark.parametrize("cons", CONS_CASES)
def test_constraint_layout_matches_oracle(cons):
    n = 24
    assert reduce_constraint_sets(cons) == orc.reduce_constraint_sets(cons)
    assert constraint_lookup_dict(reduce_constraint_sets(cons)) == orc.constraint_lookup_dict(
        orc.reduce_constraint_sets(cons))
    C = make_bond_constraint_matrix(n, cons)
    assert np.array_equal(C, orc.make_bond_constraint_matrix(n, cons))
    goa, n_red = group_layout(n, cons)
    assert C.shape == (n, n_red) and np.all(C[np.arange(n), goa] == 1)
    ptr, atoms = groups_csr(goa, n_red)
    for g in range(n_red):
        assert sorted(atoms[ptr[g]:ptr[g + 1]]) == list(np.nonzero(C[:, g])[0])


def test_layout_rejects_bad_index():
    with pytest.raises(ValueError):
        group_layout(4, {frozenset([1, 9])})


def test_linearmap_constructor_and_algebra(golden):
    g = golden("g3_linearmap.npz")
    lm = LinearMap(g["mat"])
    assert lm.n_cg_sites == 5 and lm.n_fg_sites == 15
    assert np.array_equal(LinearMap([[0, 2, 3], [4]], n_fg_sites=6).standard_matrix, g["list_ctor"])
    other = LinearMap(g["other"])
    assert np.allclose((lm @ other).standard_matrix, g["matmul"], rtol=0, atol=1e-14)
    assert np.array_equal((2.5 * lm).standard_matrix, g["rmul"])
    assert np.array_equal((lm + lm).standard_matrix, g["add"])
    assert np.array_equal(lm.T.standard_matrix, g["T"])
    a32 = lm.astype(np.float32)
    assert a32.standard_matrix.dtype == np.float32 and np.array_equal(a32.standard_matrix, g["astype32"])
    with pytest.raises(ValueError):
        LinearMap(g["mat"], n_fg_sites=15)
    with pytest.raises(ValueError):
        LinearMap([[0], [1]])
    with pytest.raises(ValueError):
        LinearMap(np.array([[np.nan, 1.0]]))
    LinearMap(np.array([[np.nan, 1.0]]), handle_nans=False)
    assert LinearMap(np.eye(4)).close_to_identity() and not lm.close_to_identity()
    assert LinearMap([[0, 2, 3], [4]], n_fg_sites=6).participating_fg == [[0, 2, 3], [4]]
    with pytest.raises(ValueError):
        lm.flat_call(np.zeros((2, 15, 3)))
    with pytest.raises(ValueError):
        lm.flat_call(np.zeros((2, 44)))


def test_uni_map_matches_saved_cln025(golden):
    g = golden("g4_cln025.npz")
    cmap = LinearMap([[int(i)] for i in g["ca"]], n_fg_sites=int(g["n_atoms"]))
    tm = constraint_aware_uni_map(traj=None, coord_map=cmap, constraints=cons_from_array(g["pairs"]))
    assert ((tm.force_map.standard_matrix - g["basic"]) ** 2).sum() < 1e-5
    assert np.array_equal(make_bond_constraint_matrix(175, cons_from_array(g["pairs"])), g["con_mat"])


def test_id_feat_labels_are_the_references(golden):
    """Integer labels bit-identical to the reference's id_feat(..., return_ids=True), label ORDER
    included (featlinearmap.py:598-609): g5 and the 30 reference-generated cases of g8."""
    g = golden("g5_feat_id.npz")
    cons = cons_from_array(g["cons"])
    cmap = LinearMap(g["coord_matrix"])
    ids = id_feat(g["coords"], cmap, cons, return_ids=True)
    ref = g["ids"]
    assert ids.dtype == np.int32 and np.array_equal(ids, ref)
    g8 = golden("g8_id_labels.npz")
    assert int(g8["n_cases"]) >= 20
    for k in range(int(g8["n_cases"])):
        n = int(g8[f"c{k}__n"])
        cons_k = cons_in_insertion_order(g8[f"c{k}__cons"])
        ours = id_feat(np.zeros((1, n, 3), np.float32), LinearMap([[0]], n_fg_sites=n), cons_k, return_ids=True)
        assert ours.dtype == np.int32 and np.array_equal(ours, g8[f"c{k}__ids"]), k
        assert np.array_equal(orc.id_feat_ids(n, cons_in_insertion_order(g8[f"c{k}__cons"])), g8[f"c{k}__ids"]), k
        if f"c{k}__feats" in g8.files:
            feats = id_feat(np.zeros((2, n, 3), np.float32), LinearMap([[0]], n_fg_sites=n), cons_k)["feats"][0]
            assert feats.dtype == np.float32 and np.array_equal(feats, g8[f"c{k}__feats"]), k
    out = id_feat(g["coords"], cmap, cons)
    f = out["feats"]
    assert len(f) == 4 and f[0] is f[3] and f[0].shape == (48, 12, len(set(ids))) and f[0].dtype == np.float32
    assert np.all(f[0].sum(axis=2) == 1) and np.all(f[0][0, np.arange(12), ids] == 1)
    assert np.array_equal(constraint_group_labels(5, {frozenset([3, 1])}), orc.id_feat_ids(5, {frozenset([3, 1])}))
    z = Multifeaturize([id_feat, id_feat])(g["coords"], cmap, cons)
    assert isinstance(z, FeatZipper) and z["names"] is None
    joined = list(z["feats"])
    assert len(joined) == 4 and joined[0].shape == (48, 12, 2 * f[0].shape[2])
    assert list(z["divs"])[0].shape == (48, 2 * f[0].shape[2], 3)
    with pytest.raises(KeyError):
        z["nope"]
    assert np.array_equal(smear_map(reduce_constraint_sets(cons), 12, return_mapping_matrix=True), g["smear"])


def test_curry_helpers():
    def f(a, b, c=0, d=0):
        return (a, b, c, d)

    assert curry(f, 2, d=4)(1, c=3) == (1, 2, 3, 4)
    cu = Curry(f, 2, d=4)
    assert cu(1, c=3) == (1, 2, 3, 4) and "Kw:" in repr(cu) and "kwargs:" in str(cu)
    assert flatten([[1, 2], [3]]) == [1, 2, 3]


def test_c_abi_exports_every_declared_symbol():
    """libaggf.so loads without a GPU and exports everything include/aggf.h declares."""
    header = open(os.path.join(ROOT, "include", "aggf.h")).read()
    declared = set(re.findall(r"\b(aggf_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in aggf.h but not exported"
    assert declared == set(_lib.PROTOTYPES), "ctypes prototypes out of sync with aggf.h"
    loaded = _lib.load()
    assert loaded.aggf_version() == 100
    assert loaded.aggf_gram_workspace_bytes(1000, 256, 256, _lib.F64, _lib.F64, 0) > 0
    assert loaded.aggf_eq_qp_workspace_bytes(100, 10, 10) > 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu():
    lm = LinearMap(np.eye(3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lm(np.zeros((2, 3, 3)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        aggforce_amd.project_forces(np.zeros((4, 3, 3)), np.ones((4, 3, 3)), lm, constrained_inds=set())


# ------------------------------------------------------------------ N > 1 path on CPU (gloo)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, T, N, seed, out_dir):
    import torch.distributed as dist
    from aggforce_amd.distributed import all_reduce_sum_, frame_shard, world_size

    sys.path.insert(0, ROOT)
    from oracle import aggforce_oracle as o

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)
    forces = rng.standard_normal((T, N, 3))
    b, e = frame_shard(T, rank, world)
    # per-shard Gram from the oracle (checker) -> product's reduction path
    local = o.qp_form(forces[b:e])
    G = torch.from_numpy(local.T @ local)
    all_reduce_sum_(G, True)
    assert world_size(True) == world
    acc = torch.tensor([float((forces[b:e] ** 2).sum()), float(forces[b:e].size)], dtype=torch.float64)
    all_reduce_sum_(acc, dist.group.WORLD)
    np.save(os.path.join(out_dir, f"G{rank}.npy"), G.numpy())
    np.save(os.path.join(out_dir, f"acc{rank}.npy"), acc.numpy())
    dist.destroy_process_group()


def test_frame_sharded_gram_allreduce_gloo(tmp_path):
    import torch.multiprocessing as mp
    from aggforce_amd.distributed import frame_shard

    T, N, seed, world = 37, 6, 5, 2
    shards = [frame_shard(T, r, world) for r in range(world)]
    assert shards[0][0] == 0 and shards[-1][1] == T and shards[0][1] == shards[1][0]
    assert [frame_shard(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    mp.spawn(_shard_worker, args=(world, _free_port(), T, N, seed, str(tmp_path)), nprocs=world, join=True)
    forces = np.random.default_rng(seed).standard_normal((T, N, 3))
    full = orc.qp_form(forces)
    G_full = full.T @ full
    for r in range(world):
        G = np.load(tmp_path / f"G{r}.npy")
        assert np.allclose(G, G_full, rtol=1e-12, atol=1e-12)
        acc = np.load(tmp_path / f"acc{r}.npy")
        assert np.isclose(acc[0] / acc[1], np.mean(forces ** 2))
    # replicated ranks hold bit-identical reduced matrices -> identical replicated solves
    assert np.array_equal(np.load(tmp_path / "G0.npy"), np.load(tmp_path / "G1.npy"))


def test_load_trajectory_npz_and_npy(tmp_path):
    from aggforce_amd.stream import default_chunk_frames, load_trajectory

    rng = np.random.default_rng(0)
    c, f = rng.random((7, 5, 3)), rng.random((7, 5, 3))
    np.savez(tmp_path / "traj.npz", coords=c, Fs=f)  # the reference's file layout
    lc, lf = load_trajectory(str(tmp_path / "traj.npz"))
    assert np.array_equal(lc, c) and np.array_equal(lf, f)
    with pytest.raises(KeyError):
        load_trajectory(str(tmp_path / "traj.npz"), forces_key="forces")
    np.save(tmp_path / "x_coords.npy", c)
    np.save(tmp_path / "x_forces.npy", f)
    mc, mf = load_trajectory(str(tmp_path / "x"))
    assert isinstance(mc, np.memmap) and np.array_equal(mf, f)
    assert default_chunk_frames(4096, 8) == (8 << 30) // (4 * 4096 * 24)


def test_grid_cv_gram_reuse_applicability():
    """Which project_forces_grid_cv calls may take the one-pass Gram-reuse form (host logic only)."""
    from aggforce_amd import LinearMap, qp_linear_map
    from aggforce_amd.agg import _gram_reuse_applicable, process_cvargs
    from aggforce_amd.qp import constraint_aware_uni_map

    cmap = LinearMap([[0], [2]], n_fg_sites=4)
    base = dict(coord_map=cmap, constrained_inds=None)
    assert _gram_reuse_applicable(["l2_regularization"], base)
    assert _gram_reuse_applicable([], dict(base, l2_regularization=1.0, method=qp_linear_map))
    assert not _gram_reuse_applicable(["l2_regularization"], dict(base, constrained_inds="auto"))
    assert not _gram_reuse_applicable(["l2_regularization"], dict(coord_map=cmap))  # default is "auto"
    assert not _gram_reuse_applicable(["l2_regularization"], dict(base, method=constraint_aware_uni_map))
    assert not _gram_reuse_applicable(["var"], base)
    assert not _gram_reuse_applicable(["l2_regularization"], dict(base, kbt=1.0))
    grid = process_cvargs({"l2_regularization": [0.0, 1.0], "x": ["a"]})
    assert [g[1] for g in grid] == [{"l2_regularization": 0.0, "x": "a"}, {"l2_regularization": 1.0, "x": "a"}]
    assert grid[0][0].l2_regularization == 0.0 and grid[1][0].x == "a"


def test_grid_cv_featurised_reuse_applicability():
    """Which featurised project_forces_grid_cv calls take the one-pass form (host logic only)."""
    from aggforce_amd import LinearMap
    from aggforce_amd.agg import _feat_reuse, process_cvargs
    from aggforce_amd.qp import Multifeaturize, gb_feat, id_feat, qp_feat_linear_map
    from aggforce_amd.util import Curry

    cmap = LinearMap([[0], [2]], n_fg_sites=4)
    fused = Multifeaturize([id_feat, Curry(gb_feat, outer=8.0, n_basis=4)])
    base = dict(coord_map=cmap, constrained_inds=set(), method=qp_feat_linear_map, featurizer=fused, kbt=0.6)
    grid = process_cvargs({"l2_regularization": [1.0, 10.0]})
    assert _feat_reuse(grid, ["l2_regularization"], base) is not None
    assert _feat_reuse(process_cvargs({}), [], dict(base, n_constraint_frames=5)) is not None  # default l2 = 10
    assert _feat_reuse(process_cvargs({"l2_regularization": [0.0, 1.0]}), ["l2_regularization"], base) is None
    assert _feat_reuse(grid, ["l2_regularization"], dict(base, constrained_inds="auto")) is None
    assert _feat_reuse(grid, ["l2_regularization"], dict(base, comm=object())) is None
    assert _feat_reuse(grid, ["l2_regularization"], dict(base, fused=False)) is None
    assert _feat_reuse(process_cvargs({"kbt": [0.5]}), ["kbt"], base) is None
    generic = Multifeaturize([lambda *a, **k: None])
    assert _feat_reuse(grid, ["l2_regularization"], dict(base, featurizer=generic)) is None
    plain = dict(base)
    del plain["method"]
    assert _feat_reuse(grid, ["l2_regularization"], plain) is None  # the linear optimiser has its own one-pass form


def test_grid_cv_noised_reuse_applicability():
    from aggforce_amd import LinearMap, joptgauss_map, stagedjoptgauss_map
    from aggforce_amd.agg import _noised_reuse_applicable

    cmap = LinearMap([[0], [2]], n_fg_sites=4)
    base = dict(coord_map=cmap, constrained_inds=None, method=joptgauss_map, var=0.01, kbt=0.6)
    assert _noised_reuse_applicable(["l2_regularization"], base)
    assert _noised_reuse_applicable([], dict(base, seed=3, l2_regularization=1.0))
    assert not _noised_reuse_applicable(["var"], base)
    assert not _noised_reuse_applicable(["l2_regularization"], dict(base, constrained_inds="auto"))
    assert not _noised_reuse_applicable(["l2_regularization"], dict(base, method=stagedjoptgauss_map))
    assert _noised_reuse_applicable(["l2_regularization"], dict(base, comm=object()))  # per-rank folds, all-reduced
    assert not _noised_reuse_applicable(["l2_regularization"], dict(base, noise=[None]))
    del base["var"]
    assert not _noised_reuse_applicable(["l2_regularization"], base)


def test_staged_map_signatures_match_reference_names():
    import inspect

    import aggforce_amd as pkg

    sig = inspect.signature(pkg.stagedjoptgauss_map)
    for name in ("traj", "coord_map", "var", "kbt", "force_map", "constraints", "seed",
                 "premap_l2_regularization", "premap_solver_args"):
        assert name in sig.parameters
    sig = inspect.signature(pkg.stagedjforcegauss_map)
    assert sig.parameters["contribution_tolerance"].default == 1e-6
    sig = inspect.signature(pkg.stagedjslicegauss_map)
    assert sig.parameters["warn_input_forces"].default is True


def test_integration_stub_matches_the_abi():
    """The ctypes stub shown in INTEGRATION.md binds real symbols with the right number of arguments."""
    import re

    from aggforce_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\nimport ctypes as C, torch\n(.*?)```", text, re.S)
    assert block is not None
    code = "import ctypes as C, torch\n" + block.group(1)
    code = code.replace('C.CDLL("libaggf.so")', f'C.CDLL("{_lib.LIB_PATH}")')
    ns: dict = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)  # noqa: S102
    for name in ("aggf_gram", "aggf_gram_workspace_bytes", "aggf_eq_qp_solve", "aggf_eq_qp_workspace_bytes",
                 "aggf_linearmap_apply"):
        assert len(getattr(ns["_l"], name).argtypes) == len(_lib.PROTOTYPES[name][1]), name


def test_constraint_layout_random_sets_match_oracle():
    """The reduced-variable layout (column order of the reference's make_bond_constraint_matrix) for
    random constraint sets: overlapping pairs, chains, larger groups, duplicates."""
    rng = np.random.default_rng(20260101)
    for _ in range(300):
        n = int(rng.integers(2, 60))
        cons = set()
        for _ in range(int(rng.integers(0, n))):
            size = int(rng.choice([2, 2, 2, 3, 4]))
            if size <= n:
                cons.add(frozenset(int(i) for i in rng.choice(n, size=size, replace=False)))
        assert reduce_constraint_sets(cons) == orc.reduce_constraint_sets(cons)
        C = make_bond_constraint_matrix(n, cons)
        assert np.array_equal(C, orc.make_bond_constraint_matrix(n, cons))
        goa, n_red = group_layout(n, cons)
        assert C.shape == (n, n_red) and np.all(C[np.arange(n), goa] == 1) and C.sum() == n


def test_bench_launcher_dry_run_two_ranks():
    """`python bench.py --gpus 2` as typed (no WORLD_SIZE): the parent starts 2 fresh ranks through
    torch.distributed.run, forwards rank 0's single JSON line, and fails when a rank fails.  --dry-run keeps
    the children off the GPU (gloo; the rendezvous / barrier / max-over-ranks / replicated-result skeleton)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    bench = os.path.join(ROOT, "bench.py")
    ok = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], capture_output=True, env=env, timeout=300)
    assert ok.returncode == 0, ok.stderr.decode()[-2000:]
    lines = [l for l in ok.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size_seen"] == 2 and rec["replicated_solve_max_abs_diff"] == 0.0
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run", "--dry-run-fail-rank", "1"],
                         capture_output=True, env=env, timeout=300)
    assert bad.returncode != 0 and not bad.stdout.strip()
    # the launcher branch never imports torch (no GPU initialisation in the parent)
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.argv=['bench.py']; import bench; "
                            "assert 'torch' not in sys.modules; print('clean')"], capture_output=True, env=env,
                           cwd=ROOT, timeout=120)
    assert probe.stdout.decode().strip() == "clean", probe.stderr.decode()[-2000:]


def _helpers_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from aggforce_amd.distributed import (agree_on_indices, agree_on_min, all_reduce_minmax_, frame_shard,
                                          shard_extent, take_global_frames, world_size)

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    T = 23
    full = torch.arange(T * 4 * 3, dtype=torch.float32).reshape(T, 4, 3) * 0.5 - 7.0
    b, e = frame_shard(T, rank, world)
    local = full[b:e].clone()
    first, total = shard_extent(local.shape[0], True, local.device)
    assert (first, total) == (b, T) and world_size(True) == world
    # an unseeded draw differs per rank: rank 0's wins everywhere
    mine = np.random.default_rng(100 + rank).choice(T, size=7, replace=False)
    idx = agree_on_indices(mine, True, local.device)
    got = take_global_frames(local, idx, True)
    lo = torch.tensor([float(rank), 5.0 - rank])
    hi = lo.clone()
    all_reduce_minmax_(lo, hi, dist.group.WORLD)
    np.save(os.path.join(out_dir, f"idx{rank}.npy"), idx)
    np.save(os.path.join(out_dir, f"got{rank}.npy"), got.numpy())
    np.save(os.path.join(out_dir, f"mm{rank}.npy"), torch.stack([lo, hi]).numpy())
    # a per-rank estimate that fixes the shape of later collectives (sites per batch of the featurised fit)
    assert agree_on_min(7 - 3 * rank, True, local.device) == 4 and agree_on_min(5, None, local.device) == 5
    try:
        take_global_frames(local, np.array([T]), True)
        raise SystemExit("expected IndexError")
    except IndexError:
        pass
    dist.destroy_process_group()


def test_sharded_frame_helpers_gloo(tmp_path):
    """The product's own multi-rank plumbing of the featurised fit on CPU tensors (gloo, 2 ranks): shard extents,
    rank 0's frame draw on every rank, frames fetched from whichever rank owns them (bit for bit), min/max reduce."""
    import torch.multiprocessing as mp

    world, T = 2, 23
    mp.spawn(_helpers_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    full = (np.arange(T * 4 * 3, dtype=np.float32).reshape(T, 4, 3) * 0.5 - 7.0)
    i0, i1 = np.load(tmp_path / "idx0.npy"), np.load(tmp_path / "idx1.npy")
    assert np.array_equal(i0, i1) and np.array_equal(i0, np.random.default_rng(100).choice(T, size=7, replace=False))
    g0, g1 = np.load(tmp_path / "got0.npy"), np.load(tmp_path / "got1.npy")
    assert np.array_equal(g0, g1) and np.array_equal(g0, full[i0])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"mm{r}.npy"), [[0.0, 4.0], [1.0, 5.0]])


def test_linear_map_survives_pickling():
    """Users keep fitted maps by pickling them (the reference's maps are plain Python objects)."""
    import pickle

    from aggforce_amd import LinearMap

    lm = LinearMap([[0, 2, 3], [4]], n_fg_sites=6, handle_nans="safe", nan_check_threshold=1e-5)
    lm.tags = {"note": "kept"} if hasattr(lm, "tags") else None
    back = pickle.loads(pickle.dumps(lm))
    assert np.array_equal(back.standard_matrix, lm.standard_matrix) and back.handle_nans == "safe"
    assert back.nan_check_threshold == 1e-5 and back.n_cg_sites == 2 and back.n_fg_sites == 6
    assert back._dev_cache == {} and back._host_ready is None


def test_solve_batches_group_sites_of_similar_size():
    """The featurised fit's batches (gbfeat._solve_batches): every site exactly once, largest first, no batch
    above the memory bound, sizes within a batch close once the batch is full enough -- and a pure function
    of its arguments (the ranks of a sharded fit must form the same batches)."""
    from aggforce_amd.qp.gbfeat import _solve_batches

    rng = np.random.default_rng(5)
    n_act = [int(x) for x in rng.integers(1255, 3050, size=64)]
    b = _solve_batches(n_act, 64, min_sites=16)
    assert sorted(i for g in b for i in g) == list(range(64))
    assert b == _solve_batches(list(n_act), 64, min_sites=16)
    assert [len(g) for g in _solve_batches(n_act, 64)] == [32, 32]  # the default: at least 32 problems per launch
    assert all(len(g) >= 16 for g in b) and len(b) >= 3
    tops = [max(n_act[i] for i in g) for g in b]
    assert tops == sorted(tops, reverse=True)
    pad = lambda n: -(-n // 64) * 64
    padded = sum(len(g) * pad(t) ** 3 for g, t in zip(b, tops))
    assert padded < 0.7 * 64 * pad(max(n_act)) ** 3  # the point of it
    # the memory bound wins over the minimum size
    b3 = _solve_batches(n_act, 3)
    assert max(len(g) for g in b3) == 3 and sorted(i for g in b3 for i in g) == list(range(64))
    # equal sizes: one batch; one site: one batch
    assert _solve_batches([500] * 10, 64) == [list(range(10))]
    assert _solve_batches([7], 1) == [[0]]


def test_host_sanitizer_build_of_the_c_abi():
    """`make asan` (sanitizers on the CPU build): the host pass of every HIP source under ASan + UBSan,
    driven through argument validation, launch planning and workspace arithmetic of all entry points by
    aggforce_amd/csrc/asan_driver.cpp -- no GPU needed (every launch fails cleanly without a device)."""
    import shutil
    import subprocess

    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    csrc = os.path.join(ROOT, "aggforce_amd", "csrc")
    run = subprocess.run(["make", "-C", csrc, "asan", "-j", "8"], capture_output=True, text=True, timeout=900)
    tail = (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0, tail
    assert "0 unexpected statuses" in run.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail


def test_no_test_module_defines_a_name_twice():
    """A second `def test_x` in one module rebinds the name and pytest silently collects only the last one (round 4:
    the stage-size cases of the few-sites apply never ran).  Every top-level function / class name of every test
    module, and every method name inside a test class, must be unique."""
    import ast
    import glob

    dups = []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "*.py"))):
        tree = ast.parse(open(path).read(), path)
        scopes = [("", tree.body)] + [(n.name + ".", n.body) for n in tree.body if isinstance(n, ast.ClassDef)]
        for prefix, body in scopes:
            seen = {}
            for node in body:
                if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                    if node.name in seen:
                        dups.append(f"{os.path.basename(path)}: {prefix}{node.name} at lines {seen[node.name]} and {node.lineno}")
                    seen[node.name] = node.lineno
    assert not dups, "duplicate definitions shadow tests:\n" + "\n".join(dups)


def test_routing_header_is_generated_from_the_committed_table():
    """make_plan's thresholds (which kernel a system takes) are constants generated from profiles/r05_routing.json, the
    table tools/routing_sweep.py measures -- not numbers typed into the planner."""
    import subprocess

    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_routing.py"), "--check"], stdout=subprocess.PIPE)
    assert rc.returncode == 0, rc.stdout.decode()
    src = open(os.path.join(ROOT, "aggforce_amd", "csrc", "aggf_gram.hip")).read()
    import json

    for name in json.load(open(os.path.join(ROOT, "profiles", "r05_routing.json")))["thresholds"]:
        assert f"routing::{name}" in src, name
