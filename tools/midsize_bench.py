"""GPU box: project_forces end to end on a mid-size molecule off the BASELINE grid -- 304 atoms (Trp-cage's count),
20 sites, float32 trajectory of 2e6 frames, bond-pair constraints on every third atom pair -- with the stage timers."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from aggforce_amd import LinearMap, project_forces  # noqa: E402
from aggforce_amd import _kernels as K  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 304
    n_cg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
    dt = torch.float64 if "f64" in sys.argv else torch.float32
    forces = K.synth_normal(T, N, dt, 1, sigma=30.0)
    coords = K.synth_normal(T, N, dt, 2, sigma=0.3, lattice=1.5)
    cmap = LinearMap([[i * (N // n_cg)] for i in range(n_cg)], n_fg_sites=N)
    cons = {frozenset([3 * i + 1, 3 * i + 2]) for i in range(N // 3)} if "nocons" not in sys.argv else set()
    out = None
    for rep in range(3):
        out = None
        if rep == 2:
            K.start_timers()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = project_forces(coords=coords, forces=forces, coord_map=cmap, constrained_inds=cons)
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
    st = K.stop_timers()
    print(json.dumps({"atoms": N, "sites": n_cg, "frames": T, "dtype": str(dt), "constraint_pairs": len(cons),
                      "ms": round(dtm * 1e3, 2), "frames_per_s": round(T / dtm),
                      "stages_ms": {k: round(v["ms"], 2) for k, v in st.items()},
                      "GB_forces": round(forces.numel() * forces.element_size() / 1e9, 2)}))


if __name__ == "__main__":
    main()
