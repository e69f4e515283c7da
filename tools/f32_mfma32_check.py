"""K1 float32 on v_mfma_f32_32x32x2_f32 (AGGF_GRAM_F32_MFMA=32) against the float64 Gram: run once per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aggforce_amd import _kernels as K
worst = 0.0
for T, N in [(37, 130), (300, 300), (2051, 385), (64, 640), (1001, 1024), (5000, 2176)]:
    f = K.synth_normal(T, N, torch.float32, T + N, sigma=30.0)
    G32 = K.gram(f, None, None, N, torch.float32)
    G64 = K.gram(f, None, None, N, torch.float64)
    err = float((G32 - G64).abs().max() / G64.abs().max())
    worst = max(worst, err)
    assert torch.equal(G32, G32.T)
print("AGGF_GRAM_F32_MFMA =", os.environ.get("AGGF_GRAM_F32_MFMA"), "worst rel err vs float64 products:", worst)
assert worst < 1e-5
