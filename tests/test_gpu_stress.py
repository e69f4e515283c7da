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
