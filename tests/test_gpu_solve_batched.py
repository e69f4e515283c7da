"""GPU: the batched equality-QP solve (one launch per factorisation step over all cg sites of the
featurised fit, featlinearmap.py:349-384) against separate solves and against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from aggforce_amd import _kernels as K  # noqa: E402
from oracle import aggforce_oracle as orc  # noqa: E402


def _problems(p, n, m, nrhs, seed, redundant=False):
    rng = np.random.default_rng(seed)
    Gs, As, Bs = [], [], []
    for _ in range(p):
        R = rng.standard_normal((3 * n + 5, n)) * rng.uniform(1, 50)
        Gs.append(R.T @ R)
        A = rng.standard_normal((m, n))
        if redundant:  # consistent but dependent rows, as the sampled constraint rows of the featurised fit
            A[m // 2:] = A[: m - m // 2] * 1.0
        As.append(A)
        x0 = rng.standard_normal((n, nrhs))
        Bs.append(A @ x0)
    return np.stack(Gs), np.stack(As), np.stack(Bs)


@pytest.mark.parametrize("p,n,m,nrhs", [(1, 40, 3, 3), (5, 70, 9, 1), (3, 300, 20, 2), (7, 64, 64, 1), (2, 513, 130, 4)])
def test_batched_solve_equals_separate_solves_and_oracle(p, n, m, nrhs):
    G, A, B = _problems(p, n, m, nrhs, seed=p * 1000 + n)
    Gd, Ad, Bd = (torch.from_numpy(x).cuda() for x in (G, A, B))
    X, st = K.eq_qp_solve_batched(Gd, 0.5, None, Ad, Bd, schur_reg=0.0, n_refine=1)
    assert X.shape == (p, nrhs, n) and st.shape == (p, 4)
    for j in range(p):
        Xj, sj = K.eq_qp_solve(Gd[j].contiguous(), 0.5, None, Ad[j].contiguous(), Bd[j].contiguous(), schur_reg=0.0,
                               n_refine=1)
        assert torch.equal(X[j], Xj) and torch.equal(st[j], sj)  # bit-identical: same kernels, same order
        ref = orc.eq_qp_solve(G[j] + 0.5 * np.eye(n), None, A[j], B[j])
        got = X[j].cpu().numpy().T
        assert np.max(np.abs(got - ref)) < 1e-8 * max(1.0, np.max(np.abs(ref)))
        assert float(st[j, 0]) == 0.0 and float(st[j, 1]) < 1e-9


def test_batched_solve_redundant_rows_and_identity_rhs():
    """Redundant sampled rows (schur_reg > 0, three refinement steps) and B = None (identity right-hand sides)."""
    p, n, m = 4, 90, 12
    G, A, B = _problems(p, n, m, 1, seed=11, redundant=True)
    Gd, Ad, Bd = (torch.from_numpy(x).cuda() for x in (G, A, B))
    X, st = K.eq_qp_solve_batched(Gd, 10.0, None, Ad, Bd, schur_reg=1e-12, n_refine=3)
    for j in range(p):
        Xj, sj = K.eq_qp_solve(Gd[j].contiguous(), 10.0, None, Ad[j].contiguous(), Bd[j].contiguous(), schur_reg=1e-12,
                               n_refine=3)
        assert torch.equal(X[j], Xj) and torch.equal(st[j], sj)
        assert float(st[j, 0]) == 0.0 and float(st[j, 1]) < 1e-8
        ref = orc.eq_qp_solve(G[j] + 10.0 * np.eye(n), None, A[j], B[j][:, 0])
        assert np.max(np.abs(X[j, 0].cpu().numpy() - ref)) < 1e-6 * max(1.0, np.max(np.abs(ref)))
    G2, A2, _ = _problems(3, 50, 6, 6, seed=12)
    X2, st2 = K.eq_qp_solve_batched(torch.from_numpy(G2).cuda(), 0.0, None, torch.from_numpy(A2).cuda(), None)
    for j in range(3):
        assert np.max(np.abs(A2[j] @ X2[j].cpu().numpy().T - np.eye(6))) < 1e-9
    # a non-positive pivot in ONE problem is reported for that problem only
    G3 = G2.copy()
    G3[1] = -G3[1]
    _, st3 = K.eq_qp_solve_batched(torch.from_numpy(G3).cuda(), 0.0, None, torch.from_numpy(A2).cuda(), None)
    st3 = st3.cpu().numpy()
    assert st3[0, 0] == 0 and st3[2, 0] == 0 and st3[1, 0] != 0
    with pytest.raises(ValueError):
        K.eq_qp_solve_batched(torch.from_numpy(G2).cuda(), 0.0, None, torch.from_numpy(A2[:2]).cuda(), None)


def test_general_solve_matches_oracle_across_block_counts():
    """The blocked factorisation behind K2 (64-wide diagonal blocks, 256-wide outer panels, inverted diagonal blocks)
    at sizes with 1, 4 and 16 diagonal blocks and ragged padding, against the oracle's exact equality-QP solve."""
    rng = np.random.default_rng(3)
    for n, m in ((64, 3), (200, 17), (1000, 64)):
        R = rng.standard_normal((3 * n, n))
        Gh = R.T @ R
        Ah = rng.standard_normal((m, n))
        X, stats = K.eq_qp_solve(torch.from_numpy(Gh).cuda(), 1e-3, None, torch.from_numpy(Ah).cuda(),
                                 torch.from_numpy(np.eye(m)).cuda(), schur_reg=1e-12, n_refine=3)
        X = X.cpu().numpy()
        assert stats.cpu().numpy()[0] == 0
        assert np.max(np.abs(Ah @ X.T - np.eye(m))) < 1e-9
        for i in (0, m - 1):
            e = np.zeros(m)
            e[i] = 1.0
            ref = orc.eq_qp_solve(Gh + 1e-3 * np.eye(n), None, Ah, e)
            assert np.max(np.abs(X[i] - ref)) < 1e-7 * max(1.0, np.max(np.abs(ref))), (n, m, i)


def test_factorisation_forms_agree(monkeypatch):
    """K2's factorisation takes one launch per 64-column step (every workgroup factors the diagonal block itself and
    walks its share of the row blocks) or, when a workgroup would have more than 8 row blocks to walk, three launches
    per step (diagonal block, panel, right-looking inner update).  ``AGGF_SOLVE_WGS=1`` (workgroups per problem of the
    step kernel; read per call) puts a 1000-variable system through BOTH: the first outer panels take the three-launch
    form, the later ones the one-launch form with ONE workgroup walking all row blocks.  Same solution as the default
    schedule (one workgroup per row block) up to rounding, same reported pivot for an indefinite matrix."""
    rng = np.random.default_rng(5)
    n, m = 1000, 40
    R = rng.standard_normal((3 * n, n))
    G = torch.from_numpy(R.T @ R).cuda()
    A = torch.from_numpy(rng.standard_normal((m, n))).cuda()
    B = torch.from_numpy(np.eye(m)).cuda()
    pins = torch.arange(0, n, n // m, dtype=torch.int32)[:m].cuda()

    def solves():
        X, st = K.eq_qp_solve(G, 1e-3, None, A, B, schur_reg=1e-12, n_refine=2)
        Xb, stb = K.eq_qp_solve_batched(torch.stack([G, 2.0 * G, G]), 1e-3, None, torch.stack([A, A, A]),
                                        torch.stack([B, B, B]), schur_reg=1e-12, n_refine=2)
        Xp, stp = K.eq_qp_solve_pinned(G, 1e-3, None, pins)
        return [X, Xb, Xp], [st, stb, stp]

    monkeypatch.delenv("AGGF_SOLVE_WGS", raising=False)
    ref, ref_st = solves()
    monkeypatch.setenv("AGGF_SOLVE_WGS", "1")
    got, got_st = solves()
    for a, b, sa, sb in zip(ref, got, ref_st, got_st):
        assert float(sa.reshape(-1, 4)[:, 0].abs().max()) == 0.0 and float(sb.reshape(-1, 4)[:, 0].abs().max()) == 0.0
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) < 1e-9 * scale
    assert float((A @ got[0].T - B).abs().max()) < 1e-9
    # an indefinite matrix: the first non-positive pivot is the same 1-based index in every form
    Gbad = G.clone()
    Gbad[700:, :] = 0.0
    Gbad[:, 700:] = 0.0
    Gbad[700:, 700:] = -torch.eye(n - 700, dtype=torch.float64, device="cuda")
    for env in (None, "1", "3"):
        if env is None:
            monkeypatch.delenv("AGGF_SOLVE_WGS", raising=False)
        else:
            monkeypatch.setenv("AGGF_SOLVE_WGS", env)
        _, st = K.eq_qp_solve(Gbad, 0.0, None, A[:, :] * 0.0 + torch.eye(m, n, dtype=torch.float64, device="cuda"), B,
                              schur_reg=0.0, n_refine=0)
        assert float(st[0]) == 701.0, (env, st)


def test_one_launch_step_is_bit_identical_for_any_workgroup_count(monkeypatch):
    """The one-launch factorisation step (chol_step_kernel) must not depend on WHICH workgroup handles a row block or
    on how many run at once: a 500-variable system stays in the one-launch form whether one workgroup walks all row
    blocks (AGGF_SOLVE_WGS=1, strictly serial) or every row block has its own (default; the form that raced in round 4:
    workgroup 0 overwrote the diagonal block late workgroups still read).  Same arithmetic per row block -> the solved
    maps, the statistics and the pinned solve are BIT-identical; a dependence on dispatch order shows as a difference.
    The replicated solve of the multi-GPU path (qp/qplinear.py:79-86 on every rank) relies on exactly this."""
    rng = np.random.default_rng(17)
    n, m = 500, 12
    R = rng.standard_normal((3 * n, n))
    G = torch.from_numpy(R.T @ R).cuda()
    A = torch.from_numpy(rng.standard_normal((m, n))).cuda()
    pins = torch.arange(0, n, n // m, dtype=torch.int32)[:m].cuda()

    def solves():
        X, st = K.eq_qp_solve(G, 1e-3, None, A, None, schur_reg=0.0, n_refine=1)
        Xb, stb = K.eq_qp_solve_batched(torch.stack([G, 3.0 * G]), 1e-3, None, torch.stack([A, A]), None)
        Xp, stp = K.eq_qp_solve_pinned(G, 1e-3, None, pins)
        return [X, st, Xb, stb, Xp, stp]

    results = {}
    for wgs in (None, "1", "2", "5"):
        if wgs is None:
            monkeypatch.delenv("AGGF_SOLVE_WGS", raising=False)
        else:
            monkeypatch.setenv("AGGF_SOLVE_WGS", wgs)
        results[wgs] = solves()
    for wgs in ("1", "2", "5"):
        for a, b in zip(results[None], results[wgs]):
            assert torch.equal(a, b), f"AGGF_SOLVE_WGS={wgs} changes the result"
    ref = orc.eq_qp_solve(G.cpu().numpy() + 1e-3 * np.eye(n), None, A.cpu().numpy(), np.eye(m))
    assert np.max(np.abs(results[None][0].cpu().numpy().T - ref)) < 1e-8 * np.max(np.abs(ref))
