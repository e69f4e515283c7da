"""GPU: the fused tile kernel of K1 (aggf_gram_gather) -- constraint-group sums, float32 -> float64 conversion and
padding to 128-column tiles inside the MFMA operand read, no packed copy -- against the packed-copy pipeline of
aggf_gram (the default; the fused kernel is opt-in through AGGF_GRAM_GATHER=1 because it measured slower, see
_kernels.gram_gather_ok; same products, so agreement to rounding of the split-K sums) and against the CPU oracle's
`qp_form(F) @ con_mat` Gram matrix (qplinear.py:66-71)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from aggforce_amd import LinearMap, Trajectory, qp_linear_map  # noqa: E402
from aggforce_amd import _kernels as K  # noqa: E402
from aggforce_amd.qp.qplinear import LinearProblem  # noqa: E402
from oracle import aggforce_oracle as orc  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def constraints(kind, N, rng):
    if kind == "none":
        return set()
    if kind == "pairs":
        return {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
    if kind == "mixed":  # pairs, triples, quads (heavy atom + hydrogens), some members a few atoms away
        cons, a = set(), 0
        while a + 6 < N:
            size = int(rng.integers(1, 5))
            if size > 1:
                members = [a] + [a + int(k) for k in rng.choice(np.arange(1, 6), size=size - 1, replace=False)]
                cons.add(frozenset(members))
            a += 6
        return cons
    raise KeyError(kind)


CASES = [
    # (T, N, in dtype, gram dtype, constraints)
    (300, 600, np.float64, np.float64, "pairs"),
    (257, 600, np.float64, np.float64, "mixed"),       # ragged frame count (stages of 4)
    (1001, 300, np.float32, np.float64, "none"),       # the reference's arithmetic for float32 forces; N % 128 != 0
    (640, 700, np.float32, np.float64, "mixed"),
    (800, 500, np.float32, np.float32, "pairs"),       # float32 products (stages of 8 frames)
    (64, 1300, np.float64, np.float64, "pairs"),       # few frames, many tiles
    (5000, 1024, np.float32, np.float64, "none"),      # whole tiles, conversion only
]


@pytest.mark.parametrize("T,N,in_dt,g_dt,kind", CASES)
def test_fused_tile_kernel_matches_packed_pipeline_and_oracle(T, N, in_dt, g_dt, kind, monkeypatch):
    rng = np.random.default_rng(N + T)
    forces = (30 * rng.standard_normal((T, N, 3))).astype(in_dt)
    cons = constraints(kind, N, rng)
    cmap = LinearMap([[0], [7]], n_fg_sites=N)
    prob = LinearProblem(cmap, cons, torch.device("cuda"))
    fd = torch.from_numpy(forces).cuda()
    taken = {"n": 0}
    real = K.gram_gather
    monkeypatch.setattr(K, "gram_gather", lambda *a, **k: (taken.__setitem__("n", taken["n"] + 1), real(*a, **k))[1])
    monkeypatch.delenv("AGGF_GRAM_GATHER", raising=False)
    Gp = prob.gram(fd, g_dt)                                   # default: packed copy + panel kernel
    assert taken["n"] == 0
    monkeypatch.setenv("AGGF_GRAM_GATHER", "1")                # opt-in: the fused kernel
    G = prob.gram(fd, g_dt)
    assert taken["n"] == 1
    tol = 1e-12 if g_dt == np.float64 else 2e-5
    assert torch.equal(G, G.T) and rel(G.cpu().numpy(), Gp.cpu().numpy()) < tol
    pr = orc.linear_problem(forces, np.asarray(cmap.standard_matrix), cons, 0.0)
    assert G.shape == pr["qp_mat"].shape and rel(G.cpu().numpy(), pr["qp_mat"]) < (1e-12 if g_dt == np.float64 else 2e-5)
    # accumulate over two frame blocks == one call; two runs are bit-identical
    half = (T // 8) * 4
    if half:
        H = prob.gram(fd[:half].contiguous(), g_dt)
        prob.gram(fd[half:].contiguous(), g_dt, out=H, accumulate=True)
        assert rel(H.cpu().numpy(), G.cpu().numpy()) < tol
    assert torch.equal(prob.gram(fd, g_dt), G)


def test_fused_tile_kernel_fallbacks_and_end_to_end(monkeypatch):
    rng = np.random.default_rng(3)
    T, N = 200, 400
    forces = 30 * rng.standard_normal((T, N, 3))
    taken = {"n": 0}
    real = K.gram_gather
    monkeypatch.setattr(K, "gram_gather", lambda *a, **k: (taken.__setitem__("n", taken["n"] + 1), real(*a, **k))[1])
    monkeypatch.setenv("AGGF_GRAM_GATHER", "1")
    # a group of five members: the packed pipeline; members far apart (window too wide for two LDS stages): too
    big = {frozenset([0, 1, 2, 3, 4])} | {frozenset([3 * i + 10, 3 * i + 11]) for i in range(100)}
    far = {frozenset([5, N - 1])} | {frozenset([3 * i + 10, 3 * i + 11]) for i in range(100)}
    cmat = orc.list_mapping_matrix([[8], [200]], N)
    for cons in (big, far):
        W = qp_linear_map(Trajectory(coords=forces, forces=forces), LinearMap(cmat), cons).force_map.standard_matrix
        assert rel(W, orc.qp_linear_map(forces, cmat, cons)) < 1e-9
    assert taken["n"] == 0
    # odd byte count (T * 3N * 4 not a multiple of 16): the packed pipeline
    f32 = forces[:3, :399].astype(np.float32)
    prob = LinearProblem(LinearMap([[0]], n_fg_sites=399), set(), torch.device("cuda"))
    prob.gram(torch.from_numpy(f32).cuda(), np.float64)
    assert taken["n"] == 0
    # end to end through qp_linear_map with pair constraints (n_red 300): fused kernel, pinned solve
    cons = {frozenset([4 * i, 4 * i + 1]) for i in range(100)}
    W = qp_linear_map(Trajectory(coords=forces, forces=forces), LinearMap(cmat), cons, 0.2).force_map.standard_matrix
    assert taken["n"] == 1 and rel(W, orc.qp_linear_map(forces, cmat, cons, 0.2)) < 1e-9
    # NaN / inf in the trajectory propagate exactly as through the packed copy (no 0 * NaN from absent members)
    bad = forces.copy()
    bad[17, 4, 1] = np.nan
    prob = LinearProblem(LinearMap(cmat), cons, torch.device("cuda"))
    Gb = prob.gram(torch.from_numpy(bad).cuda()).cpu().numpy()
    col = prob.goa[4]
    assert np.isnan(Gb[col]).all() and np.isnan(Gb[:, col]).all()
    assert np.isfinite(np.delete(np.delete(Gb, col, 0), col, 1)).all()
