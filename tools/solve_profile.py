"""GPU box: run the K2 solve alone (n = 4096, m = 256) a few times -- for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aggforce_amd import _kernels as K

n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 256)
rng = np.random.default_rng(0)
B = rng.normal(size=(n + 64, n))
G = torch.from_numpy(B.T @ B).cuda()
A = np.zeros((m, n)); A[np.arange(m), np.arange(m) * (n // m)] = 1.0
A = torch.from_numpy(A).cuda()
sizes = torch.ones(n, dtype=torch.float64, device="cuda")
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    X, stats = K.eq_qp_solve(G, 0.0, sizes, A)
    torch.cuda.synchronize()
    print(f"solve {1e3 * (time.perf_counter() - t0):.2f} ms, stats {stats.cpu().numpy()}")
