"""GPU box: re-measure the crossovers behind profiles/r05_routing.json.  For every routing rule of aggf_gram's
make_plan both routes are forced (AGGF_GRAM_ROUTE=stream | tile; AGGF_GRAM_PACK=serial | overlap for the pack rule) on
systems around the threshold, ~6 GB of frames each; the table's `measurements` are replaced, the crossovers printed,
and with --accept written into `thresholds` (then run tools/gen_routing.py and rebuild).

    python tools/routing_sweep.py [--accept] [--only rule1,rule2] [out.json]

--only: re-measure just these rules (the other rules' measurements and thresholds stay as they are).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from aggforce_amd import _kernels as K  # noqa: E402
from aggforce_amd import _lib  # noqa: E402
from aggforce_amd.constraints import group_layout, groups_csr  # noqa: E402

TABLE = os.path.join(ROOT, "profiles", "r05_routing.json")


def timed(fn, n=4):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def system(n_red, pairs, sdt):
    """(frames tensor, group CSR or None, N): `pairs` = bond pairs {3i, 3i+1} on as many atoms as give n_red columns."""
    if not pairs:
        N = n_red
    else:
        N = n_red + n_red // 2  # N - N // 3 = n_red for N = 1.5 n_red
        while N - N // 3 < n_red:
            N += 1
        while N - N // 3 > n_red:
            N -= 1
    es = 8 if sdt == torch.float64 else 4
    T = int(6e9 / (3 * es * N)) // 64 * 64
    f = K.synth_normal(T, N, sdt, 11, sigma=30.0)
    gp = ga = None
    if pairs:
        cons = {frozenset([3 * i, 3 * i + 1]) for i in range(N // 3)}
        goa, nr = group_layout(N, cons)
        assert nr == n_red, (nr, n_red)
        p, a = groups_csr(goa, nr)
        gp, ga = torch.from_numpy(p).cuda(), torch.from_numpy(a).cuda()
    return f, gp, ga, N, T


RULES = [
    # (threshold, storage dtype, product dtype, pairs, reduced-column counts)
    ("stream_edge3_max_cols", torch.float64, torch.float64, False, [264, 288, 304, 320, 336, 352, 376]),
    ("stream_pack4_max_cols_f64", torch.float64, torch.float64, True, [400, 432, 464, 480, 496, 512]),
    ("stream_pack4_max_cols_f32", torch.float32, torch.float32, True, [400, 448, 480, 496, 512]),
    ("stream_edge4_max_cols_f64", torch.float64, torch.float64, False, [392, 400, 424, 448, 472, 504]),
    ("stream_edge4_max_cols_f32", torch.float32, torch.float32, False, [392, 424, 456, 480, 504]),
    # float32 frames with float64 products, read in place by the widening tile kernel
    ("stream_edge3_max_cols_widen", torch.float32, torch.float64, False, [264, 288, 304, 320, 336, 352, 376]),
    ("stream_edge4_max_cols_widen", torch.float32, torch.float64, False, [392, 400, 424, 448, 472, 504]),
]


def main():
    accept = "--accept" in sys.argv
    out = next((a for a in sys.argv[1:] if a.endswith(".json")), TABLE)
    table = json.load(open(TABLE))
    only = next((a.split("=", 1)[1] if "=" in a else sys.argv[sys.argv.index(a) + 1] for a in sys.argv[1:] if a.startswith("--only")), None)
    only = set(only.split(",")) if only else None
    meas, proposed = [], {}
    if only:
        meas = [r for r in table.get("measurements", []) if r["rule"] not in only]
    for name, sdt, cdt, pairs, cols in RULES:
        if only and name not in only:
            continue
        best_stream = None
        for n_red in cols:
            f, gp, ga, N, T = system(n_red, pairs, sdt)
            row = {"rule": name, "n_red": n_red, "atoms": N, "frames": T, "pairs": pairs,
                   "dtypes": f"{str(sdt).split('.')[1]}->{str(cdt).split('.')[1]}"}
            for route in ("stream", "tile"):
                os.environ["AGGF_GRAM_ROUTE"] = route
                _lib.load().aggf_coverage_reset()
                row[route + "_ms"] = round(timed(lambda: K.gram(f, gp, ga, n_red, cdt)), 3)
                row[route + "_kernel"] = sorted(p_.split("(")[0].replace("void aggf::", "") for p_, c in _lib.coverage(names=True).values()
                                                if c > 0 and ("gram_small" in p_ or "gram_tile" in p_ or "pack_" in p_))
            os.environ.pop("AGGF_GRAM_ROUTE", None)
            if row["stream_ms"] <= row["tile_ms"]:
                best_stream = n_red
            meas.append(row)
            print(json.dumps(row), flush=True)
            del f
        proposed[name] = best_stream if best_stream is not None else cols[0] - 1
    # pack overlap: serial against overlapped pack + tile pipeline by padded width
    over = None
    for n_red in (() if only and "pack_overlap_min_pad" not in only else (860, 1100, 1400, 1800, 2200, 2731)):
        f, gp, ga, N, T = system(n_red, True, torch.float64)
        row = {"rule": "pack_overlap_min_pad", "n_red": n_red, "n_pad": (n_red + 127) // 128 * 128, "atoms": N, "frames": T, "pairs": True,
               "dtypes": "float64->float64"}
        os.environ["AGGF_GRAM_PACK_MIN_FRAMES"] = "8192"  # (lets the overlapped form run below the threshold too)
        for form in ("serial", "overlap"):
            os.environ["AGGF_GRAM_PACK"] = form
            row[form + "_ms"] = round(timed(lambda: K.gram(f, gp, ga, n_red, torch.float64)), 3)
        os.environ.pop("AGGF_GRAM_PACK", None)
        os.environ.pop("AGGF_GRAM_PACK_MIN_FRAMES", None)
        if row["overlap_ms"] < row["serial_ms"] and over is None:
            over = row["n_pad"] - 128
        meas.append(row)
        print(json.dumps(row), flush=True)
        del f
    if not only or "pack_overlap_min_pad" in only:
        proposed["pack_overlap_min_pad"] = over if over is not None else table["thresholds"]["pack_overlap_min_pad"]["value"]
    table["measurements"] = meas
    table["proposed_by_last_sweep"] = {**table.get("proposed_by_last_sweep", {}), **proposed} if only else proposed
    for k, v in proposed.items():
        cur = table["thresholds"][k]["value"]
        print(f"{k}: table {cur}, sweep proposes {v}")
        if accept:
            table["thresholds"][k]["value"] = int(v)
    json.dump(table, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
