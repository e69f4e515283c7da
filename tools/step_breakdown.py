"""GPU box: wall-clock breakdown of one project_forces step at the bench workload (host + device)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aggforce_amd import LinearMap, Trajectory, qp_linear_map
from aggforce_amd import _kernels as K
from aggforce_amd.agg import force_smoothness
T, N, n_cg = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 4096, 256
forces = K.synth_normal(T, N, torch.float64, 42100, sigma=30.0)
coords = K.synth_normal(T, N, torch.float64, 42101, sigma=0.3, lattice=1.5)
cmap = LinearMap([[i * (N // n_cg)] for i in range(n_cg)], n_fg_sites=N)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = sync(); traj = Trajectory(coords=coords, forces=forces)
    tm = qp_linear_map(traj=traj, coord_map=cmap, constraints=set()); t1 = sync()
    mc = tm.coord_map(coords); t2 = sync()
    mf = tm.force_map(forces); t3 = sync()
    r = force_smoothness(mf); t4 = sync()
    print(f"rep {rep}: fit {1e3*(t1-t0):.1f} ms | coord map {1e3*(t2-t1):.1f} | force map {1e3*(t3-t2):.1f} | residual {1e3*(t4-t3):.1f} | total {1e3*(t4-t0):.1f}")
# inside the fit
import aggforce_amd.qp.qplinear as ql
for name in ("gram", "eq_qp_solve", "expand_map"):
    f = getattr(K, name)
    def wrap(f=f, name=name):
        def g(*a, **k):
            t = sync(); out = f(*a, **k); print(f"   {name}: {1e3*(sync()-t):.1f} ms"); return out
        return g
    setattr(K, name, wrap())
t0 = sync(); tm = qp_linear_map(traj=traj, coord_map=cmap, constraints=set()); print(f"fit total {1e3*(sync()-t0):.1f} ms")
