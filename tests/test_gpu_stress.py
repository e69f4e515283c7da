"""GPU parity: seeded random sweep of project_forces (linear path) against the oracle -- random sizes,
mapping kinds, constraint sets, dtypes, regularisation; infeasible problems must be refused."""
import importlib.util
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_tool():
    spec = importlib.util.spec_from_file_location("stress_parity", os.path.join(ROOT, "tools", "stress_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [11, 12])
def test_random_sweep_matches_oracle(seed, monkeypatch, capsys):
    sp = _load_tool()
    monkeypatch.setattr(sys, "argv", ["stress_parity.py", "30", str(seed)])
    sp.main()  # exits non-zero (SystemExit) on the first mismatch
    assert "30 cases ok" in capsys.readouterr().out


def test_infeasible_constraints_are_refused():
    from aggforce_amd import LinearMap, project_forces

    rng = np.random.default_rng(0)
    f = rng.normal(size=(60, 8, 3))
    # two CG sites inside one rigid group share a coefficient: (M C) x = e_i cannot hold for both
    with pytest.raises(ValueError, match="cannot be met|not positive definite"):
        project_forces(f, f, LinearMap([[0], [1]], n_fg_sites=8), {frozenset([0, 1])}, l2_regularization=0.1)


def test_random_apply_sweep_matches_oracle(monkeypatch, capsys):
    """LinearMap.__call__: slice/dense/sparse maps, dtype promotion, NaN policy (identical raises)."""
    spec = importlib.util.spec_from_file_location("stress_apply", os.path.join(ROOT, "tools", "stress_apply.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["stress_apply.py", "80", "5"])
    mod.main()
    assert "80 apply cases ok" in capsys.readouterr().out


def test_random_k6_and_k5_sweeps_match_oracle(monkeypatch, capsys):
    """guess_pairwise_constraints (K6) and the CondNormal augmentation (K5) on random sizes, dtypes and premaps."""
    spec = importlib.util.spec_from_file_location("stress_misc", os.path.join(ROOT, "tools", "stress_misc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(17)
    mod.sweep_k6(rng, 40)
    mod.sweep_k5(rng, 40)
    out = capsys.readouterr().out
    assert "40 K6 cases ok" in out and "40 K5 cases ok" in out


def test_bench_contract_on_tiny_workload():
    """bench.py prints exactly one JSON line on stdout with the fields the driver reads."""
    import json
    import subprocess

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2",
                          "--warmup", "1", "--cpu-frames", "64"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_random_pinned_solve_and_noised_fit_sweeps():
    """Round 3's new paths on random shapes: aggf_eq_qp_solve_pinned against the general solve and the oracle
    (sizes around the 64 / 256 panel edges, group-size weighted l2, pins anywhere), and the noised fit without the
    extended arrays against the oracle's concatenating restatement (slice and dense premaps, constraints, both
    dtypes, N and n_cg multiples of 128)."""
    import torch

    from aggforce_amd import LinearMap, Trajectory, joptgauss_map
    from aggforce_amd import _kernels as K
    from oracle import aggforce_oracle as orc

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))

    rng = np.random.default_rng(2024)
    worst = 0.0
    for case in range(24):
        n = int(rng.choice([65, 127, 128, 129, 200, 255, 257, 320, 511, 513, 700]))
        m = int(rng.integers(1, max(2, min(n - 1, 130))))
        l2 = float(rng.choice([0.0, 0.0, 0.3, 10.0]))
        F = rng.standard_normal((3 * n + 20, n)) * rng.uniform(0.5, 80, size=n)
        Gh = F.T @ F
        pins = rng.choice(n, size=m, replace=False).astype(np.int32)
        sizes = rng.integers(1, 5, size=n).astype(np.float64)
        A = np.zeros((m, n))
        A[np.arange(m), pins] = 1.0
        Xp, sp = K.eq_qp_solve_pinned(torch.from_numpy(Gh).cuda(), l2, torch.from_numpy(sizes).cuda(),
                                      torch.from_numpy(pins).cuda())
        assert sp.cpu().numpy()[0] == 0, (case, n, m)
        Xo = orc.eq_qp_solve(Gh + l2 * np.diag(sizes), None, A, np.eye(m)).T
        worst = max(worst, rel(Xp.cpu().numpy(), Xo))
        assert np.array_equal(Xp.cpu().numpy()[:, pins], np.eye(m))
    assert worst < 1e-8, worst
    print(f"24 pinned-solve cases ok (worst rel {worst:.1e})")
    for case in range(6):
        N = int(rng.choice([128, 256, 384]))
        n_cg = 128
        dt = rng.choice([np.float32, np.float64])
        T = int(rng.integers(N + 150, N + 400))
        coords = (4 * rng.random((T, N, 3))).astype(dt)
        forces = (25 * rng.standard_normal((T, N, 3))).astype(dt)
        if rng.random() < 0.5:
            cmat = orc.list_mapping_matrix([[int(i)] for i in rng.choice(N, size=n_cg, replace=False)], N)
        else:
            cmat = orc.list_mapping_matrix([[int(i) for i in rng.choice(N, size=int(rng.integers(1, 4)), replace=False)]
                                            for _ in range(n_cg)], N)
            if np.linalg.matrix_rank(cmat) < n_cg:
                continue
        cons = {frozenset(int(i) for i in rng.choice(N, size=2, replace=False)) for _ in range(int(rng.integers(0, N // 8)))}
        l2 = float(rng.choice([0.0, 0.5]))
        var = float(rng.choice([0.01, 0.2]))
        eps = [rng.standard_normal((T, n_cg, 3)).astype(np.float32) for _ in range(2)]
        try:
            o = orc.joptgauss_force_map(coords, forces, cmat, var, 0.7, eps[0], cons, l2,
                                        dtype=np.float32 if dt == np.float32 else np.float64)
        except np.linalg.LinAlgError:
            continue
        tm = joptgauss_map(Trajectory(coords=coords, forces=forces), LinearMap(cmat), var=var, kbt=0.7, constraints=cons,
                           noise=list(eps), l2_regularization=l2)
        tol = 2e-3 if dt == np.float32 else 5e-5
        W = tm.tmap.force_map.standard_matrix
        assert rel(W, o["force_map"]) < tol, (case, N, dt, rel(W, o["force_map"]))
        fc, ff = orc.augment(coords, forces, cmat, var, 0.7, eps[1], dtype=np.float32 if dt == np.float32 else np.float64)
        mapped = tm(Trajectory(coords=coords, forces=forces))
        assert rel(mapped.forces, orc.linearmap_apply(ff, o["force_map"])) < tol
        assert rel(mapped.coords, fc[:, N:, :]) < 1e-5
    print("noised-fit cases ok")


def test_random_batched_solve_with_sparse_rows_sweep():
    """aggf_eq_qp_solve_batched_shift on random batches: caller-formed A'A, a permutation that takes the variables the
    (sparse) constraint rows touch last, and the restricted forward solve -- against the oracle's KKT solve, problem by
    problem.  Sizes straddle the 64 / 256 block edges; redundant rows (rank-deficient A) as the featurised fit has them."""
    import torch

    from aggforce_amd import _kernels as K
    from oracle import aggforce_oracle as orc

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))

    rng = np.random.default_rng(404)
    worst = 0.0
    for case in range(14):
        p = int(rng.integers(1, 5))
        n = int(rng.choice([130, 256, 300, 511, 513, 640, 900]))
        nt = int(rng.integers(8, max(9, n // 3)))          # variables the rows may touch
        m = int(rng.integers(2, 2 * nt))                   # more rows than touched variables: redundant rows
        l2 = float(rng.choice([0.5, 3.0, 40.0]))
        Gs, As, bs, perms = [], [], [], []
        for q in range(p):
            R = rng.standard_normal((n + 50, n)) * rng.uniform(0.5, 20, size=n)
            Gs.append(R.T @ R)
            t = np.sort(rng.choice(n, size=nt - int(rng.integers(0, 4)), replace=False))
            A = np.zeros((m, n))
            basis = rng.standard_normal((min(m, max(1, len(t) // 2)), len(t)))    # rank <= len(t) / 2
            A[:, t] = rng.standard_normal((m, basis.shape[0])) @ basis
            x_feas = rng.standard_normal(n)
            As.append(A)
            bs.append((A @ x_feas)[:, None])                                       # consistent right-hand side
            mask = np.zeros(n, dtype=bool)
            mask[t] = True
            perms.append(np.concatenate([np.nonzero(~mask)[0], np.nonzero(mask)[0]]).astype(np.int32))
        G_d, A_d, b_d = (torch.from_numpy(np.stack(x)).cuda() for x in (Gs, As, bs))
        AtA = torch.tril(A_d.transpose(1, 2) @ A_d).contiguous()
        perm = torch.from_numpy(np.stack(perms)).cuda()
        X, st = K.eq_qp_solve_batched(G_d, l2, None, A_d, b_d, schur_reg=1e-12, n_refine=3, AtA=AtA, perm=perm,
                                      a_first_col=n - nt)
        st = st.cpu().numpy()
        assert np.all(st[:, 0] == 0) and np.all(st[:, 1] < 1e-8), (case, st)
        for q in range(p):
            xo = orc.eq_qp_solve(Gs[q] + l2 * np.eye(n), None, As[q], bs[q])[:, 0]
            worst = max(worst, rel(X[q, 0].cpu().numpy(), xo))
    assert worst < 1e-7, worst
    print(f"14 batched sparse-row solve cases ok (worst rel {worst:.1e})")
