"""Generate golden fixtures under tests/golden/ by running the REFERENCE itself.

TEST INFRASTRUCTURE.  Runs only in the build container, where /root/reference is
mounted: it puts ``/root/reference/src`` and the stand-in ``qpsolvers`` module
(``oracle/standin``; our exact equality-QP solve in place of OSQP) on sys.path,
imports ``aggforce`` read-only (no bytecode is written), feeds it small seeded
inputs and writes inputs + the reference's outputs as .npz fixtures.  Every
fixture is also compared with the NumPy restatement in
``oracle/aggforce_oracle.py`` right here, so a mismatch fails generation.

The reference's source never enters this repository; fixtures are data only
(arrays), including the reference's own test data arrays (water dimer forces,
CLN025 saved force matrices).

    python oracle/gen_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
if not os.path.isdir(os.path.join(REF, "src", "aggforce")):
    raise SystemExit("gen_golden.py needs the reference mounted at /root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle", "standin"))
sys.path.insert(0, os.path.join(REF, "src"))

import numpy as np  # noqa: E402

import aggforce  # noqa: E402
from aggforce import LinearMap, project_forces, guess_pairwise_constraints  # noqa: E402
from aggforce.qp import (  # noqa: E402
    qp_linear_map,
    constraint_aware_uni_map,
    make_bond_constraint_matrix,
    qp_form,
    id_feat,
    qp_feat_linear_map,
    Multifeaturize,
)
import aggforce.qp.featlinearmap as ref_flm  # noqa: E402
from aggforce.trajectory import Trajectory, AugmentedTrajectory, Augmenter  # noqa: E402
from aggforce.trajectory.simplegausstraj import SimpleCondNormal  # noqa: E402
from aggforce.map import AugmentedTMap, lmap_augvariables, smear_map  # noqa: E402
from aggforce.constraints import reduce_constraint_sets  # noqa: E402

from oracle import aggforce_oracle as orc  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
SEED = 42100


def cons_to_array(cons):
    """Constraints -> (n,2) int array of pairs / padded lists for storage."""
    rows = [sorted(c) for c in cons]
    width = max([len(r) for r in rows], default=0)
    arr = -np.ones((len(rows), max(width, 1)), dtype=np.int64)
    for i, r in enumerate(rows):
        arr[i, : len(r)] = r
    return arr


def check(name, a, b, rtol=1e-9, atol=1e-9):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(1.0, float(np.max(np.abs(a))) if a.size else 1.0)
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    if not err <= atol * scale + rtol * scale:
        raise SystemExit(f"ORACLE MISMATCH {name}: max abs err {err:.3e} (scale {scale:.3e})")
    print(f"  oracle == reference  {name:40s} err {err:.2e}")


def run_linear(coords, forces, cmat, cons, l2, handle_nans=True):
    cmap = LinearMap(cmat, handle_nans=handle_nans)
    res = project_forces(coords=coords, forces=forces, coord_map=cmap, constrained_inds=cons,
                         l2_regularization=l2)
    W = res["tmap"].force_map.standard_matrix
    con_mat = make_bond_constraint_matrix(cmat.shape[1], cons)
    reg_mat = np.matmul(qp_form(forces), con_mat)
    qp_mat = reg_mat.T @ reg_mat
    if l2 > 0:
        qp_mat = qp_mat + l2 * con_mat.T @ con_mat
    return {"W": W, "mapped_forces": res["mapped_forces"], "mapped_coords": res["mapped_coords"],
            "residual": np.float64(res["residual"]), "con_mat": con_mat, "qp_mat": qp_mat,
            "A": cmat @ con_mat}


# ---------------------------------------------------------------- G1 water dimer
def g1():
    print("G1 water dimer")
    d = np.load(os.path.join(REF, "tests/data/waterdimer.npz"))
    coords, forces = d["coords"], d["Fs"]
    cmat = orc.list_mapping_matrix([[0], [3]], 6)
    guessed = guess_pairwise_constraints(coords[:10])
    check("guess_pairwise_constraints", cons_to_array(guessed),
          cons_to_array(orc.guess_pairwise_constraints(coords[:10])))
    out = {"coords": coords, "forces": forces, "coord_matrix": cmat,
           "guessed_constraints": cons_to_array(guessed)}
    for cname, cons in (("none", set()), ("guessed", guessed)):
        for l2 in (0.0, 1.0, 1e3):
            r = run_linear(coords, forces, cmat, cons, l2)
            o = orc.project_forces(coords, forces, cmat, cons, l2)
            key = f"{cname}_l2_{l2:g}"
            check(f"W {key}", r["W"], o["force_map"], rtol=1e-8)
            check(f"mapped_forces {key}", r["mapped_forces"], o["mapped_forces"], rtol=1e-8)
            pr = orc.linear_problem(forces, cmat, cons, l2)
            check(f"qp_mat {key}", r["qp_mat"], pr["qp_mat"])
            for k in ("W", "mapped_forces", "residual", "qp_mat", "A", "con_mat"):
                out[f"{key}__{k}"] = r[k]
    # the reference's own known answer for this call site (tests/test_agg.py:43-44)
    out["known_answer"] = np.array([[1, 1, 1, 0, 0, 0], [0, 0, 0, 1, 1, 1]], dtype=float)
    assert np.allclose(out["none_l2_0__W"], out["known_answer"], atol=5e-3)
    np.savez_compressed(os.path.join(OUT, "g1_waterdimer.npz"), **out)


# ---------------------------------------------------------------- G2 synthetic
def g2():
    print("G2 synthetic linear")
    rng = np.random.default_rng(SEED)
    T, N, n_cg = 64, 24, 4
    forces = (30 * rng.standard_normal((T, N, 3)))
    coords = 10 * rng.random((T, N, 3))
    slice_map = orc.list_mapping_matrix([[0], [6], [12], [18]], N)
    dense_map = rng.random((n_cg, N))
    dense_map /= dense_map.sum(axis=1, keepdims=True)
    block_map = orc.list_mapping_matrix([list(range(6 * i, 6 * i + 6)) for i in range(4)], N)
    cons_sets = {
        "none": set(),
        "pairs": {frozenset([0, 1]), frozenset([6, 7]), frozenset([12, 13])},
        "chain": {frozenset([0, 1]), frozenset([1, 2]), frozenset([2, 3]), frozenset([18, 20]),
                  frozenset([20, 23])},
        "overlap": {frozenset([4, 5]), frozenset([5, 9]), frozenset([9, 4]), frozenset([10, 11, 14]),
                    frozenset([14, 15])},
    }
    out = {"coords": coords, "forces": forces}
    maps = {"slice": slice_map, "dense": dense_map, "block": block_map}
    for mname, cmat in maps.items():
        out[f"map_{mname}"] = cmat
    for cname, cons in cons_sets.items():
        out[f"cons_{cname}"] = cons_to_array(cons)
    for mname, cmat in maps.items():
        for cname, cons in cons_sets.items():
            for dt in (np.float64, np.float32):
                for l2 in (0.0, 2.5):
                    f = forces.astype(dt)
                    c = coords.astype(dt)
                    r = run_linear(c, f, cmat, cons, l2)
                    o = orc.project_forces(c, f, cmat, cons, l2)
                    key = f"{mname}_{cname}_{np.dtype(dt).name}_l2_{l2:g}"
                    check(f"W {key}", r["W"], o["force_map"], rtol=1e-7)
                    check(f"mf {key}", r["mapped_forces"], o["mapped_forces"], rtol=1e-7)
                    check(f"mc {key}", r["mapped_coords"], o["mapped_coords"])
                    assert r["mapped_forces"].dtype == np.float64  # promotion through fp64 W
                    for k in ("W", "mapped_forces", "mapped_coords", "residual", "qp_mat"):
                        out[f"{key}__{k}"] = r[k]
    np.savez_compressed(os.path.join(OUT, "g2_synthetic_linear.npz"), **out)


# ---------------------------------------------------------------- G3 LinearMap semantics
def g3():
    print("G3 LinearMap semantics")
    rng = np.random.default_rng(seed=SEED)
    pos = 100 * (rng.random(size=(20, 15, 3)) - 0.5)  # tests/test_linearmap.py:29-33
    rng = np.random.default_rng(seed=SEED)
    mat = rng.random(size=(5, 15))  # tests/test_linearmap.py:36-43
    lm = LinearMap(mat)
    out = {"pos": pos, "mat": mat, "call": lm(pos)}
    flat = pos.reshape(20, 45)
    out["flat_call"] = lm.flat_call(flat)
    out["T"] = lm.T.standard_matrix
    other = LinearMap(rng.random(size=(15, 15)))
    out["other"] = other.standard_matrix
    out["matmul"] = (lm @ other).standard_matrix
    out["rmul"] = (2.5 * lm).standard_matrix
    out["add"] = (lm + lm).standard_matrix
    out["astype32"] = lm.astype(np.float32).standard_matrix
    out["call32"] = lm.astype(np.float32)(pos.astype(np.float32))
    out["call_mixed"] = lm(pos.astype(np.float32))
    lst = LinearMap([[0, 2, 3], [4]], n_fg_sites=6)
    out["list_ctor"] = lst.standard_matrix
    check("list ctor", lst.standard_matrix, orc.list_mapping_matrix([[0, 2, 3], [4]], 6))
    check("call", out["call"], orc.linearmap_apply(pos, mat))
    # NaN policy: NaNs only where the map has zeros -> allowed
    slicemat = orc.list_mapping_matrix([[0, 1], [5], [9, 10, 11]], 15)
    nanpos = pos.copy()
    nanpos[:, [2, 3, 4, 12], :] = np.nan
    nanpos[3, 7, 1] = np.nan
    sl = LinearMap(slicemat)
    before = nanpos.copy()
    out["nan_pos"] = nanpos
    out["nan_mat"] = slicemat
    out["nan_call"] = sl(nanpos)
    assert np.array_equal(np.isnan(before), np.isnan(nanpos))  # restored
    check("nan call", out["nan_call"], orc.linearmap_apply(nanpos, slicemat))
    # dependence on a NaN -> ValueError
    bad = pos.copy()
    bad[2, 5, 0] = np.nan
    for hn in (True, "safe"):
        try:
            LinearMap(slicemat, handle_nans=hn)(bad.copy())
            raise SystemExit("expected ValueError")
        except ValueError:
            pass
    try:
        orc.linearmap_apply(bad, slicemat)
        raise SystemExit("expected ValueError")
    except ValueError:
        pass
    out["nan_bad_pos"] = bad
    # handle_nans=False: plain product, NaN propagates
    out["nan_off_call"] = LinearMap(slicemat, handle_nans=False)(bad)
    np.savez_compressed(os.path.join(OUT, "g3_linearmap.npz"), **out)


# ---------------------------------------------------------------- G4 CLN025 structure
def g4():
    print("G4 CLN025 saved maps")
    basic = np.loadtxt(os.path.join(REF, "tests/data/cln_basic_force_mat.txt"))
    opt = np.loadtxt(os.path.join(REF, "tests/data/cln_opt_force_mat.txt"))
    ca = []
    idx = 0
    with open(os.path.join(REF, "tests/data/cln025.pdb")) as fh:
        for line in fh:
            if line.startswith("ATOM"):
                if line[12:16].strip() == "CA":
                    ca.append(idx)
                idx += 1
    n = idx
    # constraint groups = sets of atoms sharing an identical column in the saved optimum
    cols = {}
    for a in range(n):
        cols.setdefault(tuple(np.round(opt[:, a], 12)), []).append(a)
    groups = [frozenset(v) for v in cols.values() if len(v) > 1]
    # express as pairwise bond constraints (anchor-member), as guess_pairwise_constraints would
    pairs = set()
    for g in groups:
        s = sorted(g)
        for m in s[1:]:
            pairs.add(frozenset([s[0], m]))
    cmap = LinearMap([[i] for i in ca], n_fg_sites=n)
    t = Trajectory(coords=np.zeros((1, n, 3)), forces=np.zeros((1, n, 3)))
    uni = constraint_aware_uni_map(traj=t, coord_map=cmap, constraints=pairs)
    W = uni.force_map.standard_matrix
    assert ((W - basic) ** 2).sum() < 1e-5, "recovered constraints do not reproduce saved basic map"
    check("uni map", W, orc.constraint_aware_uni_map(cmap.standard_matrix, pairs))
    C = make_bond_constraint_matrix(n, pairs)
    np.savez_compressed(
        os.path.join(OUT, "g4_cln025.npz"),
        basic=basic, opt=opt, ca=np.array(ca), pairs=cons_to_array(pairs), n_atoms=n,
        con_mat=C, coord_matrix=cmap.standard_matrix,
    )


# ---------------------------------------------------------------- G5 featurised (id_feat)
class _RecordingRng:
    def __init__(self, seed, log):
        self._rng = np.random.default_rng(seed)
        self._log = log

    def choice(self, *a, **k):
        r = self._rng.choice(*a, **k)
        self._log.append(np.asarray(r).copy())
        return r


def g5():
    print("G5 featurised path with id_feat")
    rng = np.random.default_rng(SEED + 5)
    T, N = 48, 12
    forces = (20 * rng.standard_normal((T, N, 3))).astype(np.float32)
    coords = (8 * rng.random((T, N, 3))).astype(np.float32)
    cons = {frozenset([2, 3]), frozenset([6, 7]), frozenset([7, 8])}
    cmat = orc.list_mapping_matrix([[0], [2], [6], [10]], N)
    cmap = LinearMap(cmat)
    ids = id_feat(coords, cmap, cons, return_ids=True)
    check("id_feat ids", ids, orc.id_feat_ids(N, cons))
    fr = id_feat(coords, cmap, cons)
    of, od = orc.id_feat(T, N, cons)
    check("id_feat feats", fr["feats"][0], of)
    log = []
    counter = [0]

    def fake_default_rng(*a, **k):
        counter[0] += 1
        return _RecordingRng(SEED + counter[0], log)

    ref_flm.default_rng = fake_default_rng
    kbt = 0.6955215
    out = {"coords": coords, "forces": forces, "coord_matrix": cmat, "cons": cons_to_array(cons),
           "ids": ids, "kbt": kbt}
    for l2 in (10.0, 0.5):
        log.clear()
        traj = Trajectory(coords=coords, forces=forces)
        tm = qp_feat_linear_map(traj=traj, coord_map=cmap, featurizer=id_feat, kbt=kbt,
                                constraints=cons, n_constraint_frames=5, l2_regularization=l2)
        coefs = tm.force_map.tags["coef_list"]
        mapped = tm(traj)
        frames = [x.copy() for x in log]
        ocoefs = orc.qp_feat_linear_map(forces, cmat, [of] * 4, [od] * 4, kbt, frames, l2)
        for i in range(4):
            check(f"feat coef l2={l2:g} site {i}", coefs[i], ocoefs[i], rtol=1e-7)
        omapped = orc.cla_apply(forces, [of] * 4, [od] * 4, ocoefs)
        check(f"feat mapped forces l2={l2:g}", mapped.forces, omapped, rtol=1e-6)
        out[f"l2_{l2:g}__coefs"] = np.stack(coefs)
        out[f"l2_{l2:g}__frames"] = np.stack(frames)
        out[f"l2_{l2:g}__mapped_forces"] = mapped.forces
        out[f"l2_{l2:g}__mapped_coords"] = mapped.coords
    # Multifeaturize of two id_feat copies exercises FeatZipper concatenation
    log.clear()
    traj = Trajectory(coords=coords, forces=forces)
    tm = qp_feat_linear_map(traj=traj, coord_map=cmap, featurizer=Multifeaturize([id_feat, id_feat]),
                            kbt=kbt, constraints=cons, n_constraint_frames=5, l2_regularization=10.0)
    out["multi__coefs"] = np.stack(tm.force_map.tags["coef_list"])
    out["multi__frames"] = np.stack([x.copy() for x in log])
    out["multi__mapped_forces"] = tm(traj).forces
    # smear matrix, used by gb_feat
    red = reduce_constraint_sets(cons)
    sm = smear_map(site_groups=red, n_sites=N, return_mapping_matrix=True)
    check("smear", sm, orc.smear_matrix(red, N))
    out["smear"] = sm
    np.savez_compressed(os.path.join(OUT, "g5_feat_id.npz"), **out)


# ---------------------------------------------------------------- G6 augmented path
class InjectedNoiseNormal(Augmenter):
    """NumPy augmenter y = Mx + sqrt(var) eps with eps supplied by the caller (test only)."""

    def __init__(self, var, M, noise_list, dtype=np.float32):
        self.var, self.M, self.noise, self.dtype = var, M, list(noise_list), dtype

    def sample(self, source):
        eps = self.noise.pop(0)
        mean = np.einsum("tfd,cf->tcd", np.asarray(source, self.dtype), self.M.astype(self.dtype))
        return (mean + self.dtype(np.sqrt(self.var)) * eps.astype(self.dtype)).astype(self.dtype)

    def log_gradient(self, source, generated):
        M = self.M.astype(self.dtype)
        mean = np.einsum("tfd,cf->tcd", np.asarray(source, self.dtype), M)
        r = (np.asarray(generated, self.dtype) - mean) / self.dtype(self.var)
        return np.einsum("tcd,cf->tfd", r, M).astype(self.dtype), (-r).astype(self.dtype)

    def astype(self, dtype, *a, **k):
        return InjectedNoiseNormal(self.var, self.M, self.noise, np.dtype(dtype).type)


def g6():
    print("G6 augmented (noised) path")
    rng = np.random.default_rng(SEED + 6)
    T, N, n_cg = 80, 10, 3
    forces = (15 * rng.standard_normal((T, N, 3))).astype(np.float32)
    coords = (5 * rng.random((T, N, 3))).astype(np.float32)
    cmat = orc.list_mapping_matrix([[0], [4], [8]], N)
    var, kbt = 0.01, 0.6955215
    eps_fit = rng.standard_normal((T, n_cg, 3)).astype(np.float32)
    eps_apply = rng.standard_normal((T, n_cg, 3)).astype(np.float32)
    cons = {frozenset([1, 2])}
    aug = InjectedNoiseNormal(var, cmat, [eps_fit, eps_apply])
    traj = Trajectory(coords=coords, forces=forces)
    at = AugmentedTrajectory.from_trajectory(t=traj, augmenter=aug, kbt=kbt)
    oc, of = orc.augment(coords, forces, cmat, var, kbt, eps_fit)
    check("aug coords", at.coords, oc, rtol=1e-6)
    check("aug forces", at.forces, of, rtol=1e-6)
    acm = lmap_augvariables(at)
    atm = qp_linear_map(traj=at, coord_map=acm, constraints=cons)
    W = atm.force_map.standard_matrix
    o = orc.joptgauss_force_map(coords, forces, cmat, var, kbt, eps_fit, cons)
    check("aug W", W, o["force_map"], rtol=1e-6)
    tmap = AugmentedTMap(aug_tmap=atm, augmenter=aug, kbt=kbt)
    mapped = tmap(traj)  # consumes eps_apply
    oc2, of2 = orc.augment(coords, forces, cmat, var, kbt, eps_apply)
    check("aug mapped forces", mapped.forces, orc.linearmap_apply(of2, W), rtol=1e-6)
    out = {"coords": coords, "forces": forces, "coord_matrix": cmat, "var": var, "kbt": kbt,
           "eps_fit": eps_fit, "eps_apply": eps_apply, "cons": cons_to_array(cons),
           "aug_coords": at.coords, "aug_forces": at.forces, "aug_coord_matrix": acm.standard_matrix,
           "W": W, "mapped_coords": mapped.coords, "mapped_forces": mapped.forces}
    # SimpleCondNormal with a seed: sample + log_gradient values (identity premap)
    s = SimpleCondNormal(var=0.3, seed=SEED)
    src = rng.standard_normal((10, 5, 3)).astype(np.float32)
    gen = s.sample(src)
    lg = s.log_gradient(src, gen)
    olg = orc.condnormal_log_gradient(src, gen, np.eye(5), 0.3)
    check("SimpleCondNormal d/dsrc", lg[0], olg[0], rtol=1e-6)
    check("SimpleCondNormal d/dgen", lg[1], olg[1], rtol=1e-6)
    out.update({"scn_src": src, "scn_gen": gen, "scn_dsrc": lg[0], "scn_dgen": lg[1]})
    np.savez_compressed(os.path.join(OUT, "g6_augmented.npz"), **out)


# ---------------------------------------------------------------- G8 id_feat label order
def ordered_cons_array(members):
    """Constraint sets IN INSERTION ORDER, members as written -> (n, w) int array padded with -1.

    The label order of the reference's id_feat depends on the hash-table layout of the
    constraint set it is handed, hence on how that set was built; tests rebuild it with
    ``build_cons`` from exactly this array."""
    width = max([len(m) for m in members], default=1)
    arr = -np.ones((len(members), max(width, 1)), dtype=np.int64)
    for i, m in enumerate(members):
        arr[i, : len(m)] = m
    return arr


def build_cons(arr):
    cons = set()
    for row in arr:
        cons.add(frozenset(int(x) for x in row if x >= 0))
    return cons


def g8():
    print("G8 id_feat labels (reference order) on random constraint sets and CLN025")
    rng = np.random.default_rng(SEED + 8)
    cases = []
    for n in (1, 2, 3, 5, 8, 8, 12, 17, 17, 24, 33, 40, 64, 64, 97, 128, 175, 200, 256, 300, 512, 1024):
        members = []
        for _ in range(int(rng.integers(0, max(1, n)))):
            size = int(rng.choice([2, 2, 2, 3, 4]))
            if n >= size:
                members.append([int(x) for x in rng.choice(n, size=size, replace=False)])
        cases.append((n, members))
    # chains (multi-round floods), a star, everything in one group, nothing constrained
    cases.append((50, [[i, i + 1] for i in range(49) if i % 7]))
    cases.append((400, [[i + 1, i] for i in range(398, -1, -1) if i % 11]))
    cases.append((30, [[0, i] for i in range(1, 30)]))
    cases.append((9, [list(range(9))]))
    cases.append((77, []))
    # the C4 bench pattern: pairs {3i, 3i+1}
    cases.append((1024, [[3 * i, 3 * i + 1] for i in range(341)]))
    # CLN025: the 59 recovered constraint groups, as anchor-member pairs (g4's set) and as whole groups
    g4d = np.load(os.path.join(OUT, "g4_cln025.npz"))
    cln_pairs = [[int(x) for x in row if x >= 0] for row in g4d["pairs"]]
    cases.append((175, cln_pairs))
    groups = {}
    for a, b in cln_pairs:
        groups.setdefault(min(a, b), {min(a, b)}).add(max(a, b))
    cases.append((175, [sorted(v) for v in groups.values()]))
    out = {"n_cases": len(cases)}
    for k, (n, members) in enumerate(cases):
        arr = ordered_cons_array(members)
        cons = build_cons(arr)
        cmap = LinearMap([[0]], n_fg_sites=n)
        ids = id_feat(np.zeros((1, n, 3), np.float32), cmap, cons, return_ids=True)
        assert ids.dtype == np.int32
        check(f"id_feat ids case {k} (n={n}, {len(members)} sets)", ids, orc.id_feat_ids(n, build_cons(arr)),
              rtol=0, atol=0)
        out[f"c{k}__n"] = n
        out[f"c{k}__cons"] = arr
        out[f"c{k}__ids"] = ids
        if n <= 24:
            feats = id_feat(np.zeros((2, n, 3), np.float32), cmap, cons)["feats"][0]
            out[f"c{k}__feats"] = feats
    np.savez_compressed(os.path.join(OUT, "g8_id_labels.npz"), **out)



if __name__ == "__main__":
    print("reference package:", os.path.dirname(aggforce.__file__))
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g8):
        if not only or fn.__name__ in only:
            fn()
    print("fixtures written to", OUT)
