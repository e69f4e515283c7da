"""GPU box: randomized sweeps of K6 (guess_pairwise_constraints) and K5 (CondNormal augmentation) against the oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from aggforce_amd import LinearMap, Trajectory, guess_pairwise_constraints  # noqa: E402
from aggforce_amd.trajectory import AugmentedTrajectory, CondNormal  # noqa: E402
from oracle import aggforce_oracle as orc  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def sweep_k6(rng, n):
    for case in range(n):
        T = int(rng.integers(2, 60))
        N = int(rng.integers(2, 150))
        dt = rng.choice([np.float32, np.float64])
        x = rng.normal(size=(T, N, 3)) * 2.0
        # rigid pairs/triples: copies of an atom displaced by a fixed vector (distance exactly constant)
        n_rigid = int(rng.integers(0, max(1, N // 4)))
        for _ in range(n_rigid):
            i, j = rng.choice(N, size=2, replace=False)
            x[:, j] = x[:, i] + rng.normal(size=3)
        x = x.astype(dt)
        thr = float(rng.choice([1e-3, 1e-2]))
        got = guess_pairwise_constraints(x, threshold=thr)
        ref = orc.guess_pairwise_constraints(x.astype(np.float64) if dt == np.float64 else x, threshold=thr)
        if got != ref:
            # borderline pairs (std within 20 % of the threshold) may fall on either side in float32
            d = orc.distances(x.astype(np.float64))
            sd = np.sqrt(np.var(d, axis=0))
            diff = got ^ ref
            bad = [p for p in diff if not (0.8 * thr < sd[tuple(sorted(p))[0], tuple(sorted(p))[1]] < 1.25 * thr)]
            if bad:
                print(f"K6 MISMATCH case {case}: T={T} N={N} {dt.__name__} thr={thr}: {sorted(map(sorted, bad))[:5]}")
                sys.exit(1)
    print(f"{n} K6 cases ok")


def sweep_k5(rng, n):
    worst = 0.0
    for case in range(n):
        T = int(rng.integers(1, 80))
        N = int(rng.integers(2, 120))
        n_cg = int(rng.integers(1, min(N, 40) + 1))
        tdt = rng.choice([np.float32, np.float64])
        kind = rng.choice(["slice", "blocks", "dense"])
        if kind == "slice":
            M = np.zeros((n_cg, N)); M[np.arange(n_cg), rng.choice(N, size=n_cg, replace=False)] = 1.0
        elif kind == "blocks":
            M = np.zeros((n_cg, N))
            for c in range(n_cg):
                idx = rng.choice(N, size=int(rng.integers(1, min(N, 6) + 1)), replace=False)
                w = rng.random(len(idx)); M[c, idx] = w / w.sum()
        else:
            M = rng.normal(size=(n_cg, N))
        var, kbt = float(rng.choice([0.002, 0.05, 0.7])), float(rng.choice([0.6, 2.5]))
        coords = rng.normal(size=(T, N, 3)).astype(tdt)
        forces = (rng.normal(size=(T, N, 3)) * 10).astype(tdt)
        eps = rng.normal(size=(T, n_cg, 3)).astype(np.float32)
        aug = CondNormal(var=var, premap=LinearMap(M), seed=1).inject_noise(eps)
        at = AugmentedTrajectory.from_trajectory(t=Trajectory(coords=coords, forces=forces), augmenter=aug, kbt=kbt)
        rc, rf = orc.augment(coords, forces, M.astype(np.float32), var, kbt, eps)
        e = max(rel(at.coords, rc), rel(at.forces, rf))
        if e > 2e-4:
            print(f"K5 MISMATCH case {case}: T={T} N={N} n_cg={n_cg} {tdt.__name__} {kind} var={var}: {e:.3e}")
            sys.exit(1)
        worst = max(worst, e)
    print(f"{n} K5 cases ok; worst relative error {worst:.2e}")


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sweep_k6(rng, n)
    sweep_k5(rng, n)
