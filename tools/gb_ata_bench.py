"""Stand-alone timing of aggf_gb_constraint_gram / aggf_gb_constraint_rows at BASELINE config 4's shapes."""
import numpy as np
import torch

from aggforce_amd import _kernels as K

rng = np.random.default_rng(0)
n_cg, G, nb, S = 64, 683, 8, 20
n_ch = G - 1
Mg = torch.from_numpy(rng.random((n_cg, G))).cuda()
gauss = torch.from_numpy(rng.random((S, n_ch, nb)).astype(np.float32)).cuda()
M2 = K.gb_group_overlap(Mg)
for n_cols in (600, 1400, 2366):
    cols = torch.from_numpy(np.sort(rng.choice(n_ch * nb, size=n_cols, replace=False)).astype(np.int32)).cuda()
    n = G + n_cols
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    A = torch.empty((S * n_cg, n), dtype=torch.float64, device="cuda")
    b = torch.empty((S * n_cg, 1), dtype=torch.float64, device="cuda")
    for name, fn in (("gram", lambda: K.gb_constraint_gram(M2, gauss, S, G, n_ch, nb, out, cols=cols)),
                     ("rows", lambda: K.gb_constraint_rows(Mg, gauss, S, G, n_ch, nb, 3, out_A=A, out_b=b, cols=cols))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"n {n} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
