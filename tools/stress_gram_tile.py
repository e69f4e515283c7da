"""GPU box: randomized sweep of aggf_gram over its routes above 256 columns: every atom count (whole panels / EDGE / rows
that are not whole 16-byte pieces: read in place by the tile kernel), the three dtype pairs (float32 frames with float64
products are widened out of LDS), random disjoint constraint groups of 2-4 atoms (or pairs only) in a third of the cases (the streaming
kernel's group sums, the pack pass in front of the tile kernel), frame blocks that start at an odd row of a larger
array (an address off the 16-byte grid) and accumulation over two blocks -- against NumPy's float64 products of the
same stored values.
    python tools/stress_gram_tile.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from aggforce_amd import _kernels as K  # noqa: E402
from aggforce_amd import _lib  # noqa: E402
from aggforce_amd.constraints import group_layout, groups_csr  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = {"f64": 0.0, "f32f64": 0.0, "f32": 0.0}
    routes = {}
    for case in range(n_cases):
        N = int(rng.integers(257, 1500))
        T = int(rng.integers(2, 3000))
        mode = str(rng.choice(["f64", "f32f64", "f32"]))
        sdt = torch.float64 if mode == "f64" else torch.float32
        cdt = torch.float32 if mode == "f32" else torch.float64
        skip = int(rng.integers(0, 2))  # 1: the block starts at row 1 of the array
        f_all = K.synth_normal(T + skip, N, sdt, int(rng.integers(1, 1 << 30)), sigma=10.0)
        f = f_all[skip:]
        # a third of the cases with constraint groups: random disjoint groups of 2-4 atoms (the pack pass above 400-512
        # reduced columns, the streaming kernel's group sums below)
        gp = ga = None
        n_red, goa = N, None
        if rng.random() < 0.33:
            perm = rng.permutation(N)
            cons, i = set(), 0
            only_pairs = rng.random() < 0.5  # (no group of more than two atoms: the streaming kernel's two-member sums)
            while i + 4 <= N and len(cons) < N // 5:
                size = 2 if only_pairs else int(rng.integers(2, 5))
                cons.add(frozenset(int(a) for a in perm[i:i + size]))
                i += size
            goa, n_red = group_layout(N, cons)
            p_h, a_h = groups_csr(goa, n_red)
            gp, ga = torch.from_numpy(p_h).cuda(), torch.from_numpy(a_h).cuda()
        _lib.load().aggf_coverage_reset()
        if rng.random() < 0.3 and T > 4:
            cut = int(rng.integers(1, T))
            G = K.gram(f[:cut], gp, ga, n_red, cdt)
            K.gram(f[cut:], gp, ga, n_red, cdt, out=G, accumulate=True)
        else:
            G = K.gram(f, gp, ga, n_red, cdt)
        names = sorted({n.split("(")[0].replace("void aggf::", "").split("<")[0] for n, c in _lib.coverage(names=True).values()
                        if c > 0 and ("gram_" in n or "pack_" in n)})
        routes[tuple(names)] = routes.get(tuple(names), 0) + 1
        x = f.double().cpu().numpy()
        if goa is not None:  # column sums over the groups, as `@ con_mat` (qplinear.py:70)
            xs = np.zeros((x.shape[0], n_red, 3))
            np.add.at(xs, (slice(None), goa), x)
            x = xs
        ref = np.zeros((n_red, n_red))
        for d in range(3):
            xd = np.ascontiguousarray(x[:, :, d])
            ref += xd.T @ xd
        g = G.cpu().numpy()
        err = float(np.abs(g - ref).max() / np.abs(ref).max())
        tol = 3e-5 if mode == "f32" else 1e-13
        if not (err < tol) or not np.array_equal(g, g.T):
            print(f"FAIL case {case}: T={T} N={N} n_red={n_red} {mode} skip={skip} err={err:.3e} kernels={names}")
            sys.exit(1)
        worst[mode] = max(worst[mode], err)
    print(f"{n_cases} Gram cases ok; worst relative errors {worst}")
    for k, v in sorted(routes.items(), key=lambda kv: -kv[1]):
        print(f"  {v:4d} x {', '.join(k)}")


if __name__ == "__main__":
    main()
