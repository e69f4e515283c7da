"""GPU box: property check of the linear path at a large reduced dimension (pack path + n = 12288 solve)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aggforce_amd import LinearMap, project_forces
from aggforce_amd import _kernels as K
from aggforce_amd.constraints import group_layout, groups_csr

T, N, n_cg = 30000, 13000, 500
forces = K.synth_normal(T, N, torch.float32, 7, sigma=30.0)
coords = K.synth_normal(T, N, torch.float32, 8, sigma=0.3, lattice=1.5)
cons = {frozenset([5 * i + 1, 5 * i + 2]) for i in range(N // 5)}          # 2600 rigid pairs
sel = np.arange(n_cg) * (N // n_cg)                                         # multiples of 26: never a constrained atom... check
cmap = LinearMap([[int(i)] for i in sel], n_fg_sites=N)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = project_forces(coords, forces, cmap, cons, gram_dtype=np.float64)
torch.cuda.synchronize(); print(f"project_forces: {time.perf_counter() - t0:.2f} s")
W = torch.from_numpy(out["tmap"].force_map.standard_matrix).cuda()
goa, n_red = group_layout(N, cons)
p, a = groups_csr(goa, n_red)
G = K.gram(forces, torch.from_numpy(p).cuda(), torch.from_numpy(a).cuda(), n_red, torch.float64)
goa_d = torch.from_numpy(goa).cuda().long()
# reduced coefficients: W is constant inside a group -> X[c, g] = W[c, any atom of g]
first_atom = torch.from_numpy(a[p[:-1]]).cuda().long()
X = W[:, first_atom]
assert torch.equal(W, X[:, goa_d]), "W must be constant on constraint groups"
selg = goa_d[torch.from_numpy(sel).cuda()]
A = torch.zeros((n_cg, n_red), dtype=torch.float64, device="cuda"); A[torch.arange(n_cg), selg] = 1
# feasibility of (M C) x = e_c
feas = torch.max(torch.abs(X @ A.T - torch.eye(n_cg, dtype=torch.float64, device="cuda"))).item()
GX = X @ G
free = torch.ones(n_red, dtype=torch.bool, device="cuda"); free[selg] = False
kkt = torch.max(torch.abs(GX[:, free])).item() / torch.max(torch.abs(GX)).item()
q = K.gram_quadform(G, X).sum().item() / (3.0 * T * n_cg)
print(f"n_red {n_red}: feasibility {feas:.2e}, KKT {kkt:.2e}, residual identity {abs(q - out['residual']) / q:.2e}")
assert feas < 1e-10 and kkt < 1e-8 and abs(q - out["residual"]) < 1e-6 * q
print("big-n check ok")
