"""Conditioning of the featurised fit (featlinearmap.py:349-384) at 20 constraint frames per site -- CPU only,
the oracle against itself.  Backs the tolerances of tests/test_gpu_feat20.py:

    python oracle/feat_conditioning.py > profiles/r03_feat_conditioning.txt

Per geometry and cg site: numerical rank of the 20 n_cg constraint rows for float64 / float32 features, the
smallest kept singular value, the condition number of the reduced Hessian Z'(R'R + l2 I)Z, and how far the EXACT
optimum's mapped forces move when
  (a) the features are rounded to float32 and everything else stays float64        ["f32 features"],
  (b) additionally the Gram matrix is formed in float32 as the reference does it     ["reference arithmetic"]
      (featlinearmap.py:361-370 on float32 arrays),
  (c) the same float32 Gram is summed in another order (two interleaved halves)      ["summation order"].
(c) is the reference's own reproducibility floor for float32 input."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))  # repo root
from oracle import aggforce_oracle as orc  # noqa: E402
from oracle.feat_cases import GEOMETRIES, KBT, L2, dense_features, geometry, numerical_rank  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def main():
    print(f"{'geometry':18s} site  n_feat rank64 rank32  s_r/s_0   cond(H_red)  f32-features  reference-arith  summation-order")
    for name in GEOMETRIES + ["box14_degenerate"]:
        coords, forces, cons, cmat, kw, frames = geometry(name)
        f64, d64 = dense_features(coords, cmat, cons, kw, np.float64)
        c32, F32 = coords.astype(np.float32), forces.astype(np.float32)
        f32, d32 = dense_features(c32, cmat, cons, kw, np.float32)
        for c in range(cmat.shape[0]):
            A64, b = orc.feat_constraint_arrays(f64[c], c, cmat, frames[c])
            A32, _ = orc.feat_constraint_arrays(f32[c], c, cmat, frames[c])
            A32 = A32.astype(np.float64)
            r64, _ = orc.feat_site_problem(forces, f64[c], d64[c], KBT, 0.0)
            n = r64.shape[1]
            I = L2 * np.eye(n)
            x64 = orc.eq_qp_solve(r64.T @ r64 + I, None, A64, b)
            rw, _ = orc.feat_site_problem(forces, f32[c].astype(np.float64), d32[c].astype(np.float64), KBT, 0.0)
            xa = orc.eq_qp_solve(rw.T @ rw + I, None, A32, b)
            rs, qs = orc.feat_site_problem(F32, f32[c], d32[c], np.float32(KBT), 0.0)      # float32 throughout
            xb = orc.eq_qp_solve(qs.astype(np.float64) + I, None, A32, b)
            q2 = (rs[::2].T @ rs[::2] + rs[1::2].T @ rs[1::2]).astype(np.float64)
            xc = orc.eq_qp_solve(q2 + I, None, A32, b)
            s = np.linalg.svd(A64, compute_uv=False)
            r = numerical_rank(A64)
            _, _, Vt = np.linalg.svd(A64, full_matrices=True)
            Z = Vt[r:].T
            ev = np.linalg.eigvalsh(Z.T @ (r64.T @ r64 + I) @ Z)
            print(f"{name:18s} {c:4d} {n:7d} {r:6d} {numerical_rank(A32):6d}  {s[r - 1] / s[0]:8.1e}  {ev[-1] / ev[0]:10.1e}"
                  f"  {rel(rw @ xa, r64 @ x64):12.1e}  {rel(rs.astype(np.float64) @ xb, r64 @ x64):15.1e}"
                  f"  {rel(rs.astype(np.float64) @ xc, rs.astype(np.float64) @ xb):15.1e}")


if __name__ == "__main__":
    main()
