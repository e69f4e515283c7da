"""GPU box: K1 (aggf_gram) and K3 (aggf_linearmap_apply) over system sizes between the single-tile kernels' range and the
BASELINE sizes -- ~12 GB of fp64 frames each, unconstrained, 1/16 of the atoms as sites.  JSON lines: time, the rate
against the roofline that bounds the size (HBM below n/8 = 12.5 flop/B, fp64 MFMA above)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from aggforce_amd import LinearMap  # noqa: E402
from aggforce_amd import _kernels as K  # noqa: E402


def timed(fn, n=4):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    args = [a for a in sys.argv[1:] if a not in ("f32", "f32f64", "pairs")]
    pairs = "pairs" in sys.argv  # bond pairs {3i, 3i+1}: the sizes are ATOM counts, n_red = N - N // 3 columns
    mode = "f32" if "f32" in sys.argv else ("f32f64" if "f32f64" in sys.argv else "f64")  # storage / product dtypes
    sdt = torch.float64 if mode == "f64" else torch.float32
    cdt = torch.float32 if mode == "f32" else torch.float64
    es = 8 if mode == "f64" else 4
    peak = 157.3e12 if mode == "f32" else 78.6e12
    sizes = [int(a) for a in args] or [144, 192, 256, 320, 384, 512, 768, 1024, 2048]
    for N in sizes:
        T = int(12e9 / (3 * es * N)) // 64 * 64
        f = K.synth_normal(T, N, sdt, 11, sigma=30.0)
        gp = ga = None
        n_red = N
        if pairs:
            from aggforce_amd.constraints import group_layout, groups_csr
            goa, n_red = group_layout(N, {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)})
            p_h, a_h = groups_csr(goa, n_red)
            gp, ga = torch.from_numpy(p_h).cuda(), torch.from_numpy(a_h).cuda()
        tg = timed(lambda: K.gram(f, gp, ga, n_red, cdt))
        flop = 3.0 * T * n_red * (n_red + 1)
        n_cg = max(1, N // 16)
        m = torch.from_numpy(np.abs(np.random.default_rng(N).standard_normal((n_cg, N))) + 0.1).to(cdt).cuda()
        ta = timed(lambda: K.linearmap_apply(f, m))
        aflop = 2.0 * T * 3 * N * n_cg
        gb = f.numel() * es / 1e9
        print(json.dumps({"dtypes": mode + (" pairs" if pairs else ""), "atoms": N, "n_red": n_red, "frames": T, "GB": round(gb, 2),
                          "gram_ms": round(tg * 1e3, 3), "gram_TFLOPs": round(flop / tg / 1e12, 1),
                          "gram_frac_mfma": round(flop / tg / peak, 3), "gram_frac_hbm": round(gb / tg / 8000, 3),
                          "sites": n_cg, "apply_ms": round(ta * 1e3, 3), "apply_frac_mfma": round(aflop / ta / peak, 3),
                          "apply_frac_hbm": round(gb / ta / 8000, 3)}), flush=True)
        del f, m


if __name__ == "__main__":
    main()
