"""K1 on one site of the featurised fit at BASELINE config 4's shape (T = 20000 frames, fp64 regression matrix in the
in-place layout) for a range of split counts (AGGF_GRAM_KSPLIT) -- run once per value, the library reads the variable
once:  for k in 0 2 3 5 7 9 11 14 18 22; do AGGF_GRAM_KSPLIT=$k python tools/gram_ksplit_bench.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aggforce_amd import _kernels as K  # noqa: E402

T = 20000
for n, lead in ((1283, 0), (2134, 0), (2134, 640), (3049, 640)):
    ld = -(-n // 128) * 128
    R3 = torch.randn((T, ld, 3), dtype=torch.float64, device="cuda")
    R3[:, n:, :] = 0
    G = torch.empty((n, n), dtype=torch.float64, device="cuda")
    for _ in range(2):
        K.gram(R3, None, None, n, torch.float64, out=G, first_col=lead)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.gram(R3, None, None, n, torch.float64, out=G, first_col=lead)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = 3.0 * T * (n * (n + 1) - lead * (lead + 1))
    print(f"ksplit {os.environ.get('AGGF_GRAM_KSPLIT', 'auto')} n {n} lead {lead}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s = {flops / ms / 1e9 / 78.6:.3f}")
    del R3, G
