"""Runs ON THE GPU BOX: finds, for every instantiation of gram_small_kernel the library contains, one system
(atoms, constraint-group size, number of groups, input dtype, product dtype) that make_plan routes to it -- by trying
systems and reading the library's launch table (aggf_coverage_dump).  Writes tests/dispatch_cases.json (the parameter
list of tests/test_gpu_dispatch_classes.py::test_gram_streaming_kernel_every_instantiation) and prints the
instantiations no candidate reached.

    python tools/find_dispatch_cases.py [out.json]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from aggforce_amd import _kernels as K  # noqa: E402
from aggforce_amd import _lib  # noqa: E402
import kernel_inventory as inv  # noqa: E402


def groups(N, size, n_groups):
    """CSR of `n_groups` chains of `size` consecutive atoms at the front, the rest singletons (column order as
    make_bond_constraint_matrix gives it is not needed here: only the launch plan is read)."""
    ptr, atoms = [0], []
    used = n_groups * size
    for a in range(used, N):       # unconstrained atoms first, in atom order
        atoms.append(a)
        ptr.append(len(atoms))
    for g in range(n_groups):
        atoms.extend(range(g * size, (g + 1) * size))
        ptr.append(len(atoms))
    return np.array(ptr, np.int32), np.array(atoms, np.int32)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "dispatch_cases.json")
    compiled = {n for n in inv.compiled_kernels(_lib.LIB_PATH) if "gram_small_kernel" in n}
    found = {}
    T = 64
    pairs = ((torch.float64, torch.float64), (torch.float32, torch.float64), (torch.float32, torch.float32))
    cands = [(N, 1, 0) for N in range(1, 531)]
    for size in (2, 3, 4, 6, 8, 12):
        for N in range(size, 1712, 3):
            for frac in (1.0, 0.5, 0.25, 0.1):
                ng = int((N // size) * frac)
                if ng >= 1 and N - ng * (size - 1) <= 520:
                    cands.append((N, size, ng))
    for ind, cd in pairs:
        for N, size, ng in cands:
            n_red = N - ng * (size - 1)
            f = torch.zeros((T, N, 3), dtype=ind, device="cuda")
            gp = ga = None
            if ng:
                p, a = groups(N, size, ng)
                gp, ga = torch.from_numpy(p).cuda(), torch.from_numpy(a).cuda()
            _lib.load().aggf_coverage_reset()
            try:
                K.gram(f, gp, ga, n_red, cd)
            except Exception as e:  # noqa: BLE001
                print("skip", N, size, ng, e)
                continue
            for mangled, (pretty, cnt) in _lib.coverage(names=True).items():
                if cnt > 0 and "gram_small_kernel" in pretty and mangled not in found:
                    found[mangled] = {"kernel": pretty.split("(")[0].replace("void ", ""), "N": N, "group_size": size,
                                      "n_groups": ng, "n_red": n_red, "in": str(ind).split(".")[1],
                                      "compute": str(cd).split(".")[1]}
        print(str(ind), str(cd), "found so far", len(found), "of", len(compiled), flush=True)
    missing = sorted(inv.demangle(compiled - set(found)).values())
    cases = sorted(found.values(), key=lambda c: c["kernel"])
    with open(out_path, "w") as fh:
        json.dump({"cases": cases, "unreached": missing}, fh, indent=1)
    print(f"{len(cases)} instantiations reached, {len(missing)} not:")
    for m in missing:
        print("  ", m.split("(")[0])


if __name__ == "__main__":
    main()
