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
