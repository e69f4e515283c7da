"""GPU: the featurised fit at the reference's default of 20 constraint frames per site
(featlinearmap.py:254,445-459), end to end against the CPU oracle.

Two regimes (`oracle/feat_conditioning.py`):

* ``feature_dtype=np.float64`` -- product and oracle evaluate the same expressions in the same arithmetic:
  coefficients within 1e-6, mapped forces within 1e-7.  The constraint rows are rank deficient by construction
  (20 n_cg rows whose id_feat part does not depend on the frame), the oracle cuts the rank by SVD, the product
  regularises the Schur complement: this is where the two treatments are confronted.
* float32 features (the reference's JAX default, and the default here) -- north_star's float32 bound of 1e-3 on
  the mapped forces against the exact (float64) optimum, provided rounding the features to float32 does not
  change the numerical rank of the constraint rows (asserted as a precondition; when it does, the EXACT
  problem is ill-posed -- the oracle's own float32 and float64 answers then differ by 1e-1, which
  tests/test_oracle_golden.py::test_exact_featurised_problem_is_ill_posed_when_float32_flips_the_rank records).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from aggforce_amd import LinearMap, Trajectory  # noqa: E402
from aggforce_amd.qp import Multifeaturize, gb_feat, id_feat, qp_feat_linear_map  # noqa: E402
from aggforce_amd.util import Curry  # noqa: E402
from oracle import aggforce_oracle as orc  # noqa: E402
from oracle.feat_cases import GEOMETRIES, KBT, L2, N_FRAMES, dense_features, geometry, numerical_rank  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def objective(forces, feats, divs, coefs, l2):
    """sum over sites of x'(R'R + l2 I)x -- what featlinearmap.py:370-381 minimises."""
    tot = 0.0
    for f, d, x in zip(feats, divs, coefs):
        reg, _ = orc.feat_site_problem(np.asarray(forces, np.float64), np.asarray(f, np.float64), np.asarray(d, np.float64),
                                       KBT, 0.0)
        r = reg @ x
        tot += float(r @ r) + l2 * float(x @ x)
    return tot


@pytest.mark.parametrize("name", GEOMETRIES)
def test_twenty_constraint_frames_float64_features_match_the_oracle(name):
    coords, forces, cons, cmat, kw, frames = geometry(name)
    cmap = LinearMap(cmat)
    feat = Multifeaturize([id_feat, Curry(gb_feat, feature_dtype=np.float64, **kw)])
    traj = Trajectory(coords=coords, forces=forces)
    tm = qp_feat_linear_map(traj, cmap, feat, KBT, constraints=cons, frame_indices=frames, l2_regularization=L2)
    assert tm.force_map.tags["fit_info"]["feature_dtype"] == "float64"
    coef = np.stack(tm.force_map.tags["coef_list"])
    feats, divs = dense_features(coords, cmat, cons, kw, np.float64)
    ocoef = np.stack(orc.qp_feat_linear_map(forces, cmat, feats, divs, KBT, frames, L2))
    assert all(len(f) == N_FRAMES for f in frames)
    assert rel(coef, ocoef) < 1e-6, rel(coef, ocoef)
    mf = tm(traj).forces
    assert rel(mf, orc.cla_apply(forces, feats, divs, list(ocoef))) < 1e-7
    # the constraint rows hold on every sampled frame of every site (20 x n_cg rows each)
    for c in range(cmat.shape[0]):
        A, b = orc.feat_constraint_arrays(feats[c], c, cmat, np.asarray(frames[c]))
        assert np.max(np.abs(A @ coef[c] - b)) < 1e-8
    # the dense (generic featuriser protocol) path of the product on the same float64 features
    dense = qp_feat_linear_map(traj, cmap, feat, KBT, constraints=cons, frame_indices=frames, l2_regularization=L2,
                               fused=False)
    assert rel(np.stack(dense.force_map.tags["coef_list"]), ocoef) < 1e-6
    assert rel(dense(traj).forces, mf) < 1e-7


@pytest.mark.parametrize("name", GEOMETRIES)
def test_twenty_constraint_frames_float32_default(name):
    coords, forces, cons, cmat, kw, frames = geometry(name)
    coords, forces = coords.astype(np.float32), forces.astype(np.float32)
    cmap = LinearMap(cmat)
    feat = Multifeaturize([id_feat, Curry(gb_feat, **kw)])
    traj = Trajectory(coords=coords, forces=forces)
    tm = qp_feat_linear_map(traj, cmap, feat, KBT, constraints=cons, frame_indices=frames, l2_regularization=L2)
    coef = np.stack(tm.force_map.tags["coef_list"])
    f32, d32 = dense_features(coords, cmat, cons, kw, np.float32)
    f64, d64 = dense_features(coords, cmat, cons, kw, np.float64)
    # precondition: float32 rounding of the features leaves the numerical rank of every site's rows alone
    for c in range(cmat.shape[0]):
        A32, _ = orc.feat_constraint_arrays(f32[c], c, cmat, np.asarray(frames[c]))
        A64, _ = orc.feat_constraint_arrays(f64[c], c, cmat, np.asarray(frames[c]))
        assert numerical_rank(A32) == numerical_rank(A64), (name, c)
    exact = orc.qp_feat_linear_map(forces.astype(np.float64), cmat, f64, d64, KBT, frames, L2)      # float64 throughout
    mf_exact = orc.cla_apply(forces.astype(np.float64), f64, d64, exact)
    ref32 = orc.qp_feat_linear_map(forces, cmat, f32, d32, KBT, frames, L2)     # the reference's arithmetic: float32 Gram
    mf_ref32 = orc.cla_apply(forces, f32, d32, ref32)
    mf = tm(traj).forces
    err = rel(mf, mf_exact)
    assert err < 1e-3, err                                   # north_star: mapped forces within 1e-3 in float32
    # ... and the product (exact Gram of the float32 regression matrix) is closer to the exact optimum than the
    # reference's own float32 arithmetic is (its float32 Gram carries rounding noise of the order of l2)
    assert err <= 1.05 * rel(mf_ref32, mf_exact) + 1e-5, (err, rel(mf_ref32, mf_exact))
    o_prod, o_exact = objective(forces, f64, d64, list(coef), L2), objective(forces, f64, d64, exact, L2)
    assert abs(o_prod / o_exact - 1.0) < 1e-5, o_prod / o_exact - 1.0
    for c in range(cmat.shape[0]):
        A, b = orc.feat_constraint_arrays(f32[c], c, cmat, np.asarray(frames[c]))
        assert np.max(np.abs(A.astype(np.float64) @ coef[c] - b)) < 1e-5   # rows rebuilt by NumPy in float32
