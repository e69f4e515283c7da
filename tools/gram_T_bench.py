"""K1 at a fixed width (n = 2134 of ld 2176, fp64, in-place layout) over the number of frames: how much of the
distance between BASELINE config 4's per-site launches (T = 20000: 0.75-0.78) and config 3 (0.86) is the launch's
length?  Prints the planner's choice through the timing only (AGGF_GRAM_KSPLIT overrides it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aggforce_amd import _kernels as K  # noqa: E402

n, lead = 2134, 0
ld = -(-n // 128) * 128
for T in (5000, 10000, 20000, 40000, 80000, 160000, 320000):
    R3 = torch.randn((T, ld, 3), dtype=torch.float64, device="cuda")
    R3[:, n:, :] = 0
    G = torch.empty((n, n), dtype=torch.float64, device="cuda")
    for _ in range(2):
        K.gram(R3, None, None, n, torch.float64, out=G, first_col=lead)
    reps = max(2, int(10 * 20000 / T))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        K.gram(R3, None, None, n, torch.float64, out=G, first_col=lead)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 3.0 * T * (n * (n + 1) - lead * (lead + 1))
    print(f"T {T}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s = {flops / ms / 1e9 / 78.6:.3f}")
    del R3, G
