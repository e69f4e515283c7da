"""K1 on inputs that are not ready-made panels: the fused tile kernel (aggf_gram_gather) against the packed-copy
pipeline (aggf_gram: pack_groups_kernel + panel kernel), HIP-event times of the whole call.
    python tools/gather_bench.py [T]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aggforce_amd import LinearMap  # noqa: E402
from aggforce_amd import _kernels as K  # noqa: E402
from aggforce_amd.qp.qplinear import LinearProblem  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
CASES = [
    ("f32 -> f64 products, N 4096, no constraints", 4096, torch.float32, np.float64, "none"),
    ("f64, N 4000 (31.25 tiles), no constraints", 4000, torch.float64, np.float64, "none"),
    ("f64, N 4096, bond pairs (n_red 2731)", 4096, torch.float64, np.float64, "pairs"),
    ("f32 -> f64 products, N 4096, bond pairs", 4096, torch.float32, np.float64, "pairs"),
    ("f32 products, N 2048, bond pairs", 2048, torch.float32, np.float32, "pairs"),
    ("f32 -> f64 products, N 1000, groups of 1-4 (CH3-like)", 1000, torch.float32, np.float64, "mixed"),
]


def cons_of(kind, N):
    if kind == "none":
        return set()
    if kind == "pairs":
        return {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
    rng = np.random.default_rng(0)
    out, a = set(), 0
    while a + 5 < N:
        size = int(rng.integers(1, 5))
        if size > 1:
            out.add(frozenset(range(a, a + size)))
        a += 5
    return out


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


for name, N, dt, gdt, kind in CASES:
    f = K.synth_normal(T, N, dt, 7, sigma=30.0)
    prob = LinearProblem(LinearMap([[0]], n_fg_sites=N), cons_of(kind, N), f.device)
    os.environ["AGGF_GRAM_GATHER"] = "1"
    lay = K.gather_layout(N, prob.n_red, prob._csr)
    ok = K.gram_gather_ok(f, prob.n_red, K.torch_dtype(gdt), lay)
    t_g = timed(lambda: prob.gram(f, gdt)) if ok else float("nan")
    os.environ["AGGF_GRAM_GATHER"] = "0"
    t_p = timed(lambda: prob.gram(f, gdt))
    flops = 3.0 * T * prob.n_red * (prob.n_red + 1)
    peak = 78.6e12 if gdt == np.float64 else 157.3e12
    print(f"{name:58s} n_red {prob.n_red:5d} span {lay['span'] if lay else -1:4d} mm {lay['mm'] if lay else 0}: fused {t_g:8.2f} ms "
          f"({flops / t_g / 1e-3 / peak:.3f})  packed {t_p:8.2f} ms ({flops / t_p / 1e-3 / peak:.3f})", flush=True)
    del f
