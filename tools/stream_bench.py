"""Throughput of the out-of-core path (aggforce_amd.stream) on host-resident .npy files.

python tools/stream_bench.py [T] [N] [n_cg] [chunk_frames]  -> one JSON line (GB/s of trajectory streamed)
"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from aggforce_amd import LinearMap  # noqa: E402
from aggforce_amd.stream import load_trajectory, project_forces_streamed  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    n_cg = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        prefix = os.path.join(d, "run")
        for name, scale in (("coords", 0.3), ("forces", 30.0)):
            mm = np.lib.format.open_memmap(f"{prefix}_{name}.npy", mode="w+", dtype=np.float32, shape=(T, N, 3))
            for b in range(0, T, 20000):
                mm[b:b + 20000] = rng.standard_normal((min(20000, T - b), N, 3), dtype=np.float32) * scale
            mm.flush()
            del mm
        coords, forces = load_trajectory(prefix)
        cmap = LinearMap([[i * (N // n_cg)] for i in range(n_cg)], n_fg_sites=N)
        project_forces_streamed(coords[:2000], forces[:2000], cmap, None)  # warm-up (library, pinned pools)
        t0 = time.perf_counter()
        out = project_forces_streamed(coords, forces, cmap, None, chunk_frames=chunk or None)
        dt = time.perf_counter() - t0
    gb = 3 * T * N * 3 * 4 / 1e9  # forces twice (fit, apply) + coords once
    print(json.dumps({"T": T, "N": N, "n_cg": n_cg, "seconds": dt, "streamed_GB": gb, "GB_per_s": gb / dt,
                      "frames_per_s": T / dt, "residual": out["residual"]}))


if __name__ == "__main__":
    main()
