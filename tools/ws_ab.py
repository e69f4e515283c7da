"""GPU box: the producer-consumer streaming Gram kernel (AGGF_GRAM_WS=1, aggf_gram_ws.h) against the library's routing
without it (AGGF_GRAM_WS=0) over system sizes and layouts -- ~12 GB of frames each.  JSON lines: both times, both
fractions of the roofline that bounds the size, and whether the two matrices agree (bit for bit is not expected: the
slab sums differ in order; the maximum relative difference is printed).

    python tools/ws_ab.py [f64|f32|f32f64] [pairs] [GB=12] atoms...
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from aggforce_amd import _kernels as K  # noqa: E402
from aggforce_amd import _lib  # noqa: E402
from aggforce_amd.constraints import group_layout, groups_csr  # noqa: E402


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
    flags = {"f32", "f32f64", "f64", "pairs"}
    args = [a for a in sys.argv[1:] if a not in flags and not a.startswith("GB=")]
    gb_target = float(next((a[3:] for a in sys.argv[1:] if a.startswith("GB=")), 12))
    mode = "f32" if "f32" in sys.argv else ("f32f64" if "f32f64" in sys.argv else "f64")
    pairs = "pairs" in sys.argv
    sdt = torch.float64 if mode == "f64" else torch.float32
    cdt = torch.float32 if mode == "f32" else torch.float64
    es = 8 if mode == "f64" else 4
    peak = 157.3e12 if mode == "f32" else 78.6e12
    for N in [int(a) for a in args]:
        T = int(gb_target * 1e9 / (3 * es * N)) // 64 * 64
        f = K.synth_normal(T, N, sdt, 11, sigma=30.0)
        gp = ga = None
        n_red = N
        if pairs:
            cons = {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
            goa, n_red = group_layout(N, cons)
            p, a = groups_csr(goa, n_red)
            gp, ga = torch.from_numpy(p).cuda(), torch.from_numpy(a).cuda()
        res = {}
        for ws in ("0", "1"):
            os.environ["AGGF_GRAM_WS"] = ws
            _lib.load().aggf_coverage_reset()
            t = timed(lambda: K.gram(f, gp, ga, n_red, cdt))
            G = K.gram(f, gp, ga, n_red, cdt)
            kern = [p_.split("(")[0].replace("void aggf::", "") for p_, c in _lib.coverage(names=True).values()
                    if c > 0 and ("gram_" in p_ or "pack_" in p_) and "reduce" not in p_ and "table" not in p_]
            res[ws] = (t, G, kern)
        flop = 3.0 * T * n_red * (n_red + 1)
        gb = f.numel() * es / 1e9
        d = float((res["0"][1] - res["1"][1]).abs().max() / res["0"][1].abs().max())
        print(json.dumps({"dtypes": mode, "atoms": N, "n_red": n_red, "pairs": pairs, "frames": T, "GB": round(gb, 2),
                          "old_ms": round(res["0"][0] * 1e3, 3), "ws_ms": round(res["1"][0] * 1e3, 3),
                          "old_frac_mfma": round(flop / res["0"][0] / peak, 3), "ws_frac_mfma": round(flop / res["1"][0] / peak, 3),
                          "old_frac_hbm": round(gb / res["0"][0] / 8000, 3), "ws_frac_hbm": round(gb / res["1"][0] / 8000, 3),
                          "max_rel_diff": d, "old_kernel": res["0"][2], "ws_kernel": res["1"][2]}), flush=True)
        del f, res


if __name__ == "__main__":
    main()
