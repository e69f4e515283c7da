"""Where the host time of one project_forces step goes (c3 geometry, few frames so that the GPU stages are short):
cProfile over 30 steps, functions sorted by their own time.  `python tools/c3_hostprof.py [frames] [atoms sites f32|f64]`
(`python tools/c3_hostprof.py 100000 1024 64 f32` = BASELINE config 2, whose step is short enough for the host to show)"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aggforce_amd import LinearMap, project_forces  # noqa: E402
from aggforce_amd import _kernels as K  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_cg = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dt = torch.float32 if len(sys.argv) > 4 and sys.argv[4] == "f32" else torch.float64
forces = K.synth_normal(T, N, dt, 1, sigma=30.0)
coords = K.synth_normal(T, N, dt, 2, sigma=0.3, lattice=1.5)
cmap = LinearMap([[i * (N // n_cg)] for i in range(n_cg)], n_fg_sites=N)
for _ in range(3):
    project_forces(coords, forces, cmap, constrained_inds=set())
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    project_forces(coords, forces, cmap, constrained_inds=set())
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
import time

torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    project_forces(coords, forces, cmap, constrained_inds=set())
torch.cuda.synchronize()
print("ms per step without the profiler: %.3f" % ((time.perf_counter() - t0) / 30 * 1e3))
print("---- who calls the synchronising tensor methods")
st.print_callers(r"method 'to' of|method 'item' of|method 'cpu' of")
