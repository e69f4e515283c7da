"""GPU box: randomized sweep of LinearMap.__call__ (K3 / K3b, NaN policy, dtype promotion) against the oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from aggforce_amd import LinearMap  # noqa: E402
from oracle import aggforce_oracle as orc  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    n_raise = 0
    for case in range(n_cases):
        T = int(rng.integers(1, 300))
        N = int(rng.integers(1, 500))
        n_cg = int(rng.integers(1, min(N, 200) + 1))
        pdt = rng.choice([np.float32, np.float64])
        kind = rng.choice(["slice", "dense", "blocks"])
        if kind == "slice":
            M = np.zeros((n_cg, N))
            M[np.arange(n_cg), rng.choice(N, size=n_cg, replace=False)] = 1.0
        elif kind == "dense":
            M = rng.normal(size=(n_cg, N))
        else:
            M = np.zeros((n_cg, N))
            for c in range(n_cg):
                idx = rng.choice(N, size=int(rng.integers(1, min(N, 9) + 1)), replace=False)
                M[c, idx] = rng.random(len(idx))
        handle = bool(rng.random() < 0.8)
        pts = rng.normal(size=(T, N, 3)).astype(pdt)
        nan_kind = rng.choice(["none", "unused", "used"], p=[0.5, 0.3, 0.2])
        if nan_kind != "none":
            cols_used = np.nonzero(np.abs(M).sum(axis=0) > 0)[0]
            cols_free = np.nonzero(np.abs(M).sum(axis=0) == 0)[0]
            pool = cols_free if (nan_kind == "unused" and len(cols_free)) else cols_used
            a = int(rng.choice(pool))
            pts[int(rng.integers(0, T)), a, int(rng.integers(0, 3))] = np.nan
        lm = LinearMap(M, handle_nans=handle)
        desc = f"case {case}: T={T} N={N} n_cg={n_cg} {pdt.__name__} {kind} handle_nans={handle} nan={nan_kind}"
        try:
            ref = orc.linearmap_apply(pts, M, handle_nans=handle)
            ref_err = None
        except ValueError as e:
            ref, ref_err = None, e
        try:
            out = lm(pts)
            err = None
        except ValueError as e:
            out, err = None, e
        if (ref_err is None) != (err is None):
            print("RAISE MISMATCH", desc, "oracle:", ref_err, "product:", err)
            sys.exit(1)
        if err is not None:
            n_raise += 1
            continue
        if out.dtype != ref.dtype or out.shape != ref.shape:
            print("DTYPE/SHAPE MISMATCH", desc, out.dtype, ref.dtype, out.shape, ref.shape)
            sys.exit(1)
        both_nan = np.isnan(out) & np.isnan(ref)
        if not np.array_equal(np.isnan(out), np.isnan(ref)):
            print("NAN PATTERN MISMATCH", desc)
            sys.exit(1)
        scale = max(1e-300, float(np.nanmax(np.abs(ref)))) if np.isfinite(ref).any() else 1.0
        e = float(np.nanmax(np.abs(np.where(both_nan, 0.0, out - ref)))) / scale if out.size else 0.0
        tol = 3e-5 if pdt == np.float32 and M.dtype == np.float32 else 1e-6 if pdt == np.float32 else 1e-10
        if e > tol:
            print("VALUE MISMATCH", desc, e)
            sys.exit(1)
        worst = max(worst, e)
    print(f"{n_cases} apply cases ok ({n_raise} raised identically); worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
