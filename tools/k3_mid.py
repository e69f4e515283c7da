"""GPU box: K3 alone on a mid-size system (default: 1e6 float32 frames x 582 atoms, 35 sites, float64 map) -- a few
launches for rocprofv3 (kernel trace / PMC passes); prints the HIP-event time of each.
    python tools/k3_mid.py [frames atoms sites f32|f64]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from aggforce_amd import _kernels as K  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 582
n_cg = int(sys.argv[3]) if len(sys.argv) > 3 else 35
dt = torch.float64 if "f64" in sys.argv else torch.float32
f = K.synth_normal(T, N, dt, 3, sigma=30.0)
m = torch.from_numpy(np.random.default_rng(1).standard_normal((n_cg, N))).cuda()
for rep in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = K.linearmap_apply(f, m)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    flop = 2.0 * 3 * T * N * n_cg
    print(f"apply {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s ({flop / ms / 1e9 / 78.6:.3f} of the fp64 peak on the real sites)  "
          f"{f.numel() * f.element_size() / ms / 1e6:.0f} GB/s", flush=True)
