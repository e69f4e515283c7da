"""GPU box: run-to-run bit reproducibility of K1 / K2 / K3 at sizes where every CU is busy (a race in the stage
barriers of these kernels would show up as a difference between two identical launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aggforce_amd import _kernels as K  # noqa: E402

ok = True
for dt, T, N in ((torch.float64, 200000, 4096), (torch.float32, 400000, 2176), (torch.float64, 30011, 1024 + 128)):
    f = K.synth_normal(T, N, dt, seed=5, sigma=30.0)
    G = [K.gram(f, None, None, N, dt) for _ in range(3)]
    same = all(torch.equal(G[0], g) for g in G[1:])
    print("K1", dt, T, N, "bit-identical" if same else "DIFFERENT", float(G[0][0, 0]))
    ok &= same
    m = K.synth_normal(1, 256 * N // 3 + 1, torch.float64, seed=9, sigma=1.0).reshape(-1)[: 256 * N].reshape(256, N).contiguous()
    if dt == torch.float32:
        outs = [K.linearmap_apply(f, m) for _ in range(3)]
    else:
        outs = [K.linearmap_apply(f, m) for _ in range(3)]
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print("K3", dt, T, N, "bit-identical" if same else "DIFFERENT")
    ok &= same
    del outs
    if dt == torch.float64 and N == 4096:
        A = torch.zeros((256, N), dtype=torch.float64, device="cuda")
        A[torch.arange(256), torch.arange(256) * 16] = 1.0
        b = torch.eye(256, dtype=torch.float64, device="cuda")
        X = [K.eq_qp_solve(G[0], 0.0, None, A, b, schur_reg=1e-12, n_refine=3)[0] for _ in range(3)]
        same = all(torch.equal(X[0], x) for x in X[1:])
        print("K2", N, "bit-identical" if same else "DIFFERENT")
        ok &= same
print("determinism ok" if ok else "NONDETERMINISTIC")
sys.exit(0 if ok else 1)
