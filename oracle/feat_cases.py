"""Test geometries of the featurised fit at 20 constraint frames per site (TEST INFRASTRUCTURE, like everything
under oracle/): shared by tests/test_gpu_feat20.py, tests/test_oracle_golden.py and oracle/feat_conditioning.py.

Every geometry is a seeded synthetic trajectory; cg sites never coincide with a (group-mean) atom position, because
at r = 0 the reference's norm gradient is NaN (jaxfeat.py:451 through jnp.linalg.norm)."""
import numpy as np

from . import aggforce_oracle as orc

KBT = 0.6955215
L2 = 10.0
N_FRAMES = 20  # featlinearmap.py:254 n_constraint_frames default


def _lattice(T, N, seed, noise=0.3):
    rng = np.random.default_rng(seed)
    side = int(np.ceil(N ** (1 / 3)))
    base = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)[:N] * 1.5
    coords = base[None] + noise * rng.standard_normal((T, N, 3))
    forces = 30 * rng.standard_normal((T, N, 3)) + 3.0 * (coords - coords.mean(axis=1, keepdims=True))
    return coords, forces


def geometry(name):
    """(coords, forces, constraints, coord matrix, gb_feat kwargs, 20 frame indices per site), float64 arrays."""
    if name in ("box14", "box14_degenerate"):
        # 14 atoms in a 6 A box, widely varying distances; two-atom sites touching pair/chain constraints.
        # "_degenerate": site 2 averages two UNCONSTRAINED atoms, so both are exactly equidistant from the site
        # (it is their midpoint), their Gaussian rows coincide in exact arithmetic and the constraint rows lose
        # rank -- a rank that float32 rounding of the features restores (see oracle/feat_conditioning.py)
        rng = np.random.default_rng(0)
        T, N = 60, 14
        coords = 6 * rng.random((T, N, 3)) + 1
        forces = 25 * rng.standard_normal((T, N, 3))
        cons = {frozenset([1, 2]), frozenset([4, 5]), frozenset([5, 6]), frozenset([10, 13])}
        site2 = [8, 9] if name == "box14_degenerate" else [8, 10]
        cmat = orc.list_mapping_matrix([[0, 1], [4, 7], site2, [12, 13]], N)
        kw = dict(outer=8.0, inner=0.0, n_basis=4, width=1.0)
    elif name == "lattice64_pairs":
        # 1.5 A lattice + 0.3 A noise (BASELINE config 4's synthetic geometry, small), bond pairs {3i, 3i+1},
        # slice map on constrained atoms (the CLN025 situation: every mapped atom is bonded to a hydrogen)
        T, N = 300, 64
        coords, forces = _lattice(T, N, 1)
        cons = {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
        cmat = orc.list_mapping_matrix([[0], [15], [30], [45], [60]], N)
        kw = dict(outer=8.0, inner=0.0, n_basis=8, width=1.0)
    elif name == "lattice125_mixed":
        # pairs and triples (CH2-like), unconstrained atoms in between, sites on constrained atoms and one
        # two-atom site across two groups
        T, N = 400, 125
        coords, forces = _lattice(T, N, 2)
        cons = {frozenset([5 * i, 5 * i + 1]) for i in range(25)} | {frozenset([5 * i, 5 * i + 2]) for i in range(0, 25, 2)}
        cmat = orc.list_mapping_matrix([[0], [25, 31], [50], [75], [100], [120]], N)
        kw = dict(outer=6.0, inner=0.5, n_basis=6, width=0.8, dist_power=1.0)
    else:
        raise KeyError(name)
    rng = np.random.default_rng(3)
    frames = [rng.choice(coords.shape[0], size=N_FRAMES, replace=False) for _ in range(cmat.shape[0])]
    return coords, forces, cons, cmat, kw, frames


GEOMETRIES = ["box14", "lattice64_pairs", "lattice125_mixed"]


def dense_features(coords, cmat, cons, kw, dtype):
    """Per-site dense [id_feat | gb_feat] features and divergences as the reference's Multifeaturize hands them to
    qp_feat_linear_map (featlinearmap.py:108-111), with gb_feat evaluated in ``dtype``."""
    T, N, _ = coords.shape
    ids = orc.id_feat_ids(N, cons)
    G = int(ids.max()) + 1
    smear = orc.smear_matrix(orc.reduce_constraint_sets(cons), N) if cons else np.eye(N, dtype=np.float32)
    cg = orc.linearmap_apply(coords, cmat)
    onehot = np.zeros((T, N, G), dtype=np.float32)
    onehot[:, np.arange(N), ids] = 1
    full = dict(dist_power=0.5)
    full.update(kw)
    feats, divs = [], []
    for c in range(cmat.shape[0]):
        gf, gd = orc.gb_feat_site(coords, cg[:, c, :], ids, smear, n_channels=G - 1, dtype=dtype, **full)
        feats.append(np.concatenate([onehot.astype(dtype), gf], axis=2))
        divs.append(np.concatenate([np.zeros((T, G, 3), dtype), gd], axis=1))
    return feats, divs


def numerical_rank(A):
    """Rank of A with the cut eq_qp_solve uses (aggforce_oracle.py: max(shape) eps s_max)."""
    s = np.linalg.svd(np.asarray(A, np.float64), compute_uv=False)
    return int(np.sum(s > max(A.shape) * np.finfo(np.float64).eps * s[0]))
